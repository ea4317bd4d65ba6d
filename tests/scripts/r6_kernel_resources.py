"""Static resources and instruction mix of every kernel of the four libraries, from `hipcc --cuda-device-only -S` dumps (build container, no GPU):

    python tests/scripts/r6_kernel_resources.py <dir with *.s> > profiles/r06_kernel_resources.txt

Per kernel: VGPRs / AGPRs / SGPRs, LDS bytes, scratch bytes and spill counts (all must be 0 for the hot kernels), the workgroup size, the waves per SIMD
the register count allows (512 VGPR+AGPR budget per SIMD lane on gfx950, granule 8), and the instruction mix of the kernel body: MFMA, VALU, v_exp /
transcendental, LDS, global / buffer memory, LDS-DMA, barriers, waitcnt -- counted over the whole kernel text (unrolled loops count once per copy).
The dumps come from tests/scripts/r6_dump_isa.sh (same flags as __graft_entry__.build())."""
import os
import re
import subprocess
import sys
from collections import OrderedDict


def demangle(names):
    """`k_name<f16, 4, 2>` from the Itanium name (GNU c++filt does not know the _Float16 / __bf16 manglings DF16_ / DF16b): the kernel's own name and
    its template arguments -- types f16 / bf16 / float, integer and bool literals -- which is all these kernels are parameterised on."""
    out = {}
    for n in names:
        m = re.match(r"_ZN(?:12_GLOBAL__N_1|\d+[a-z_]+)?(\d+)", n)
        if not m:
            out[n] = n
            continue
        ln = int(m.group(1))
        base = n[m.end():m.end() + ln]
        rest = n[m.end() + ln:]
        args = []
        if rest.startswith("I"):
            i = 1
            while i < len(rest) and rest[i] != "E":
                if rest.startswith("DF16_", i):
                    args.append("f16"); i += 5
                elif rest.startswith("DF16b", i):
                    args.append("bf16"); i += 5
                elif rest[i] == "f":
                    args.append("float"); i += 1
                elif rest[i] == "L":
                    j = rest.index("E", i)
                    lit = rest[i + 2:j]
                    args.append(("true" if lit == "1" else "false") if rest[i + 1] == "b" else lit.replace("n", "-"))
                    i = j + 1
                else:
                    args.append("?"); break
        out[n] = base + ("<" + ", ".join(args) + ">" if args else "")
    return out


def short(d):
    return d if len(d) <= 60 else d[:57] + "..."


CLASSES = OrderedDict([
    ("mfma", re.compile(r"^v_(mfma|smfmac)")),
    ("trans", re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")),
    ("valu", re.compile(r"^v_")),
    ("lds", re.compile(r"^ds_")),
    ("lds_dma", re.compile(r"^(global|buffer)_load_lds")),
    ("vmem", re.compile(r"^(global|buffer|flat|scratch)_")),
    ("salu", re.compile(r"^s_(?!waitcnt|barrier|nop|endpgm|sleep|setprio)")),
    ("barrier", re.compile(r"^s_barrier")),
    ("waitcnt", re.compile(r"^s_waitcnt")),
])


def parse(path):
    txt = open(path, errors="replace").read()
    meta = {}
    m = txt.find("amdhsa.kernels:")
    if m >= 0:
        for blk in re.split(r"\n  - ", txt[m:])[1:]:
            g = lambda k, d=0: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, d])[1]
            name = g("name", None)
            if name is None:
                continue
            meta[name] = dict(vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")), lds=int(g("group_segment_fixed_size")),
                              scratch=int(g("private_segment_fixed_size")), vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count")),
                              wg=int(g("max_flat_workgroup_size")))
    # kernel bodies: from "<name>:" to the matching ".Lfunc_end"
    mix = {}
    for name in meta:
        i = txt.find("\n" + name + ":")
        if i < 0:
            continue
        j = txt.find(".Lfunc_end", i)
        counts = {k: 0 for k in CLASSES}
        counts["scratch_total"] = counts["scratch_in_mfma_blocks"] = 0
        blk_mfma, blk_scr = 0, 0      # the basic block being read: its MFMA and scratch instruction counts
        for line in txt[i:j].splitlines() + [".LBB_end:"]:
            st = line.strip()
            if re.match(r"^\.LBB\w+:", st):
                counts["scratch_total"] += blk_scr
                if blk_mfma:
                    counts["scratch_in_mfma_blocks"] += blk_scr
                blk_mfma = blk_scr = 0
                continue
            ins = st.split(" ")[0].split("\t")[0]
            if not ins or ins[0] in ".;#/" or ins.endswith(":"):
                continue
            if ins.startswith("scratch_"):
                blk_scr += 1
            if ins.startswith(("v_mfma", "v_smfmac")):
                blk_mfma += 1
            for k, rx in CLASSES.items():
                if rx.match(ins):
                    counts[k] += 1
                    break
        mix[name] = counts
    return meta, mix


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa"
    print(__doc__.split("\n\n")[0])
    print()
    bad = []
    for f in sorted(os.listdir(d)):
        if not f.endswith(".s"):
            continue
        meta, mix = parse(os.path.join(d, f))
        if not meta:
            continue
        dm = demangle(list(meta))
        print(f"== {f[:-2]}.hip: {len(meta)} kernels")
        print(f"{'kernel':60s} {'VGPR':>4s} {'AGPR':>4s} {'SGPR':>4s} {'LDS B':>6s} {'scr':>4s} {'spill':>5s} {'WG':>4s} {'w/SIMD':>6s} | {'mfma':>5s} {'valu':>6s} {'trans':>5s} {'lds':>5s} {'dma':>4s} {'vmem':>5s} {'barr':>4s} {'wait':>5s}")
        for name, r in sorted(meta.items(), key=lambda kv: short(dm[kv[0]])):
            regs = r["vgpr"] + r["agpr"]
            gran = (max(regs, 1) + 7) // 8 * 8
            occ = min(8, 512 // gran)
            c = mix.get(name, {k: 0 for k in CLASSES})
            if r["scratch"] or r["vspill"] or r["sspill"]:
                bad.append((f[:-2], short(dm[name]), r["scratch"], r["vspill"], r["sspill"], c.get("scratch_total", 0), c.get("scratch_in_mfma_blocks", 0)))
            print(f"{short(dm[name]):60s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['lds']:6d} {r['scratch']:4d} {r['vspill'] + r['sspill']:5d} {r['wg']:4d} {occ:6d} | "
                  f"{c['mfma']:5d} {c['valu']:6d} {c['trans']:5d} {c['lds']:5d} {c['lds_dma']:4d} {c['vmem']:5d} {c['barrier']:4d} {c['waitcnt']:5d}")
        print()
    print("kernels with scratch or register spills (file, kernel, scratch bytes per lane, VGPR spills, SGPR spills, scratch instructions in the kernel, "
          "of which in a basic block that holds MFMAs -- the main loops):")
    for b in bad:
        print("   ", b)
    if not bad:
        print("    none")
    hot = [b for b in bad if b[6]]
    print("scratch traffic inside an MFMA loop:", hot if hot else "none -- every spill sits in a prologue / epilogue block")


if __name__ == "__main__":
    main()

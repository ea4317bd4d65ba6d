"""Condense the two rocprofv3 --pmc passes of run_diff_pmc.sh: per kernel (conv_mfma / attn_fwd) and grid, mean counters per launch,
MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles).  Kernel cycles come from GRBM_GUI_ACTIVE, which rocprofv3
reports SUMMED over the 8 XCDs (the raw value / duration would be a 12-16 "GHz" clock): cycles = GRBM_GUI_ACTIVE / 8, i.e. the
clock the kernel actually ran at (1.5-2.1 GHz under MFMA load, not the 2.4 GHz the TFLOP/s peak assumes)."""
import collections
import csv
import glob
import json
import os
import sys

a_dir, b_dir, out = sys.argv[1:4]
traffic_dirs = sys.argv[4:6]          # optional: the FETCH_SIZE and WRITE_SIZE passes


def rows_of(d):
    fs = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    return list(csv.DictReader(open(fs[0]))) if fs else []


acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(dict)
GEMM_CASES = ["L0 ff-in 230400x2560x320 LN+GEGLU", "L0 out-proj 230400x320x320 +residual", "L2 ff-in 14400x10240x1280 LN+GEGLU", "L1 qkv 57600x1920x640 LN"]
for rows in [rows_of(a_dir), rows_of(b_dir)] + [rows_of(d) for d in traffic_dirs]:
    # the persistent GEMM launches one grid for every shape: tell the cases of diff_kernels_one.py apart by launch order (3 each)
    gemm_ids = sorted({int(r["Dispatch_Id"]) for r in rows if "k_gemm_nt" in r["Kernel_Name"]})
    gemm_case = {d: GEMM_CASES[min(i // 3, len(GEMM_CASES) - 1)] for i, d in enumerate(gemm_ids)}
    for r in rows:
        n = r["Kernel_Name"]
        if "k_conv_mfma" not in n and "k_attn_fwd" not in n and "k_attn_bwd" not in n and "k_gemm_nt" not in n and "k_attn_short" not in n:
            continue
        kind = "conv" if "k_conv_mfma" in n else ("gemm" if "k_gemm_nt" in n else ("attn_bwd_dkv" if "bwd_dkv" in n else ("attn_bwd_dq" if "bwd_dq" in n else "attn")))
        key = kind + " " + n.split("ConvArgs")[0].split("GemmArgs")[0][-40:] + f" grid={r['Grid_Size']}"
        if "k_attn_short" in n:          # the wave-per-item kernels of the temporal attention (bench.py reads the forward's traffic by this prefix)
            key = ("attn short bwd" if "short_bwd" in n else "attn short fwd") + f" grid={r['Grid_Size']}"
        if kind == "gemm":
            key = "gemm " + gemm_case[int(r["Dispatch_Id"])]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]].add(r["Dispatch_Id"])
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            dur[key][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
res = {}
with open(out + ".csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
    for key in sorted(acc):
        per = {c: acc[key][c] / max(1, len(cnt[key][c])) for c in acc[key]}
        for c, v in sorted(per.items()):
            w.writerow([key, c, round(v, 1), len(cnt[key][c])])
        d_ns = sum(dur[key].values()) / max(1, len(dur[key])) if dur[key] else None
        cyc = per.get("GRBM_GUI_ACTIVE")
        cyc = cyc / 8.0 if cyc else cyc   # summed over the 8 XCDs
        entry = {"launches": len(cnt[key].get("SQ_WAVE_CYCLES", ())), "duration_us_in_counter_pass": round(d_ns / 1e3, 1) if d_ns else None}
        if cyc and d_ns:
            entry["clock_ghz"] = round(cyc / d_ns, 3)
        if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in per:
            entry["mfma_busy"] = round(per["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
        if cyc and "SQ_ACTIVE_INST_VALU" in per:
            entry["valu_active"] = round(per["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc), 4)
        if "SQ_WAVE_CYCLES" in per:
            wc = per["SQ_WAVE_CYCLES"]
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in per:
                    entry[c.lower() + "_frac_of_wave_cycles"] = round(per[c] / wc, 4)
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_WAVES"):
            if c in per:
                entry[c.lower()] = int(per[c])
        # HBM-side bytes per launch (KB counters): FETCH_SIZE doubled -- on gfx950 it reports half the bytes of wide coalesced
        # streaming reads (MI355X_MICROARCH.md, "HBM"), which is what these kernels' 16-byte-per-lane loads / LDS-DMA are; WRITE_SIZE raw
        if "FETCH_SIZE" in per or "WRITE_SIZE" in per:
            entry["fetch_bytes_x2"] = int(2 * 1024 * per.get("FETCH_SIZE", 0.0))
            entry["write_bytes"] = int(1024 * per.get("WRITE_SIZE", 0.0))
            entry["traffic_bytes"] = entry["fetch_bytes_x2"] + entry["write_bytes"]
            if d_ns:
                entry["traffic_tb_per_s_in_counter_pass"] = round(entry["traffic_bytes"] / d_ns / 1e3, 3)
        res[key] = entry
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])

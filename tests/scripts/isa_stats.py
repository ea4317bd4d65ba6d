"""Per-basic-block instruction mix of one kernel in a hipcc -S dump (dev tool).
usage: isa_stats.py file.s <kernel-name-substring> [min_block_len]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 60
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M)
body = m.group(2)
blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
names = ["entry"] + blocks[1::2]
codes = [blocks[0]] + blocks[2::2]
for n, c in zip(names, codes):
    ins = [l.split()[0] for l in c.split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    if len(ins) < minlen:
        continue
    print(n, "n=", len(ins), "mfma=", sum("mfma" in i for i in ins),
          "valu=", sum(i.startswith("v_") and "mfma" not in i for i in ins),
          "ds=", sum(i.startswith("ds_") for i in ins), "vmem=", sum(i.startswith(("global_", "buffer_")) for i in ins),
          "salu=", sum(i.startswith("s_") for i in ins))
    print("   ", collections.Counter(ins).most_common(30))
for k in ("num_vgpr", "num_agpr", "scratch"):
    mm = re.search(r"\.set \S*" + re.escape(key) + r"\S*\." + k + r", (\S+)", s)
    print(k, mm.group(1) if mm else None)

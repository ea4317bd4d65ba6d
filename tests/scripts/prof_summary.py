"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) totals.  usage: prof_summary.py trace.csv [top] [skip-substr] [first-substr]

A trace that holds two `k_profile_marker` launches (bench.py under GVD_BENCH_MARKERS=1 brackets its timed region with them) is cut to the
launches between them.  Otherwise first-substr (default k_conv_mfma; "" = everything): launches BEFORE the first kernel whose name contains it are left out -- a bench process
builds and fills its random-weight models first (one abs-max reduction, one randn, one fp32 -> fp16 copy per parameter: ~9 000 tiny launches,
60-90 ms), which is not part of any step and made the round-4 summaries read as "8 % torch glue".  The last lines give the share of the
counted time that is NOT one of this package's kernels (torch elementwise / copy / fill / reduce launches)."""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = sys.argv[3] if len(sys.argv) > 3 else "naive_conv"
first = sys.argv[4] if len(sys.argv) > 4 else "k_conv_mfma"
rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_profile_marker" in r["Kernel_Name"]]
if len(marks) >= 2:   # bench.py under GVD_BENCH_MARKERS: the timed region only
    print(f"(timed region only: {marks[-1] - marks[0] - 1} of {len(rows)} launches lie between the first and the last k_profile_marker)")
    rows = [r for r in rows[marks[0] + 1:marks[-1]] if "k_profile_marker" not in r["Kernel_Name"]]
    first = ""
if first:
    for i, r in enumerate(rows):
        if first in r["Kernel_Name"]:
            dropped = rows[:i]
            print(f"(left out: {len(dropped)} launches, {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in dropped) / 1e6:.1f} ms before the first {first})")
            rows = rows[i:]
            break
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if skip and skip in n:
        continue
    key = (n[:90], r["Grid_Size_X"], r["Grid_Size_Y"])
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total {tot / 1e3:.1f} ms over {sum(v[0] for v in agg.values())} launches")
byname = collections.defaultdict(float)
for k, v in agg.items():
    byname[k[0][:48]] += v[1]
print("-- by kernel name")
for k, v in sorted(byname.items(), key=lambda kv: -kv[1])[:22]:
    print(f"{v / 1e3:9.1f} ms {100 * v / tot:5.1f}%  {k}")
print("-- by (kernel, grid)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1] / 1e3:9.1f} ms n={v[0]:5d} avg={v[1] / v[0]:9.1f}us grid=({k[1]},{k[2]}) {k[0]}")
is_own = lambda n: "_GLOBAL__N_" in n or "gvd" in n or "(anonymous namespace)::k_" in n
own = sum(v[1] for k, v in agg.items() if is_own(k[0]))
oth = [(k, v) for k, v in agg.items() if not is_own(k[0])]
print(f"-- launches that are not this package's kernels: {sum(v[1] for _, v in oth) / 1e3:.1f} ms = {100 * (tot - own) / tot:.2f} % of the counted kernel time, {sum(v[0] for _, v in oth)} launches")
byn = collections.defaultdict(lambda: [0, 0.0])
for k, v in oth:
    byn[k[0][:90]][0] += v[0]
    byn[k[0][:90]][1] += v[1]
for k, v in sorted(byn.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{v[1] / 1e3:9.1f} ms n={v[0]:5d}  {k}")

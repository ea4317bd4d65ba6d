"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) totals.  usage: prof_summary.py trace.csv [top] [skip-substr]"""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = sys.argv[3] if len(sys.argv) > 3 else "naive_conv"
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

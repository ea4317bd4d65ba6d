"""Kernel timeline of a rocprofv3 kernel_trace.csv: per kernel name the mean duration and the mean idle gap before it.
usage: timeline_gaps.py trace.csv [skip_first_n_launches]"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = rows[skip:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev_end = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-40:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append((e - s) / 1e3)
    if prev_end is not None:
        gap[n].append((s - prev_end) / 1e3)
    prev_end = max(prev_end or 0, e)
tot = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"span {tot / 1e3:.2f} ms, {len(rows)} launches")
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    g = gap.get(n, [0.0])
    print(f"{n:42s} n={len(dur[n]):5d} dur {sum(dur[n]) / len(dur[n]):8.2f} us   gap-before {sum(g) / len(g):7.2f} us (max {max(g):8.1f})")

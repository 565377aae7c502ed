#!/usr/bin/env python3
"""Launch-stream occupancy of one train step from a rocprofv3 kernel trace: tools/trace_gaps.py <dir with *kernel_trace.csv> [--timeline]
Steps are delimited by the dense `adam_kernel` launch that ends each one; prints span, busy time and the largest gaps of the main queue."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
i0, i1 = ends[-3], ends[-2]
seg = rows[i0 + 1:i1 + 1]
t0 = int(seg[0]["Start_Timestamp"])
mainq = collections.Counter(r["Queue_Id"] for r in seg).most_common(1)[0][0]
main = [r for r in seg if r["Queue_Id"] == mainq]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in main)
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), a["Kernel_Name"][:36], b["Kernel_Name"][:36]) for a, b in zip(main[:-1], main[1:])]
print(f"step span {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, {len(seg)} kernels ({len(main)} on the main queue), busy {busy / 1e3:.1f} us, gaps {sum(g[0] for g in gaps) / 1e3:.1f} us")
for g in sorted(gaps, reverse=True)[:8]:
    print(f"  gap {g[0] / 1e3:6.1f} us  {g[1]} -> {g[2]}")
if "--timeline" in sys.argv:
    for r in seg:
        n = r["Kernel_Name"].replace("void ", "").replace("nrl::", "")
        n = n[:n.find("(")] if "(" in n else n
        print(f"q{r['Queue_Id']} {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {n[:90]}")

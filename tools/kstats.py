#!/usr/bin/env python3
"""Print a per-step kernel breakdown from a rocprofv3 *kernel_stats.csv (usage: kstats.py DIR STEPS)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    if ms > 0.02:
        name = r["Name"].replace("void ", "").replace("nrl::", "")
        cut = name.find(">(")
        name = name[:cut + 1] if cut > 0 else name[:name.find("(")] if "(" in name else name
        print(f"{ms:7.3f} ms/step  n={int(r['Calls']) / steps:4.1f}  {name[:110]}")
print(f"{tot:7.3f} ms/step total kernel time")

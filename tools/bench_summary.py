#!/usr/bin/env python3
"""Prints the headline fields of a bench.py JSON line (file argument)."""
import json
import sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print("value", j["value"], "ms", j["ms_per_step"], "median", j["median_ms_per_step"], "frac", r["frac"], "launch ms", r["avg_launch_ms"],
      "traffic", r["traffic"], "profile ok", (r.get("traffic_profile") or {}).get("matches_running_build"))
for k in ("forward_only", "f32_engine", "lstur", "plm", "cpu_baseline"):
    v = j.get(k)
    if isinstance(v, dict):
        v = {a: b for a, b in v.items() if a not in ("config", "roofline", "sample")} | ({"frac": v["roofline"]["frac"]} if "roofline" in v else {})
    print(k, v)

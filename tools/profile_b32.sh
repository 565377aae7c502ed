#!/bin/bash
# Runs ON THE GPU BOX: per-kernel time of the B=32 step (BASELINE configs[0]) -> gpurun_out/profiles/<tag>_b32_*
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --workload mind32 --steps 100 --warmup 20 --no-extras > $OUT/${TAG}_b32_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -f csv -d $OUT/${TAG}_b32_trace -o ${TAG}_b32 -- python $R/bench.py --workload mind32 --steps 40 --warmup 10 --no-extras > $OUT/${TAG}_b32_trace.log 2>&1
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$OUT/${TAG}_b32_trace/${TAG}_b32_kernel_stats.csv")))
nk = lambda r: r.get("Name") or r.get("kernel")
calls = lambda r: int(r.get("Calls") or r.get("calls"))
tot = lambda r: float(r.get("TotalDurationNs", 0)) / 1e6 if "TotalDurationNs" in r else float(r["total_ms"])
steps = max(calls(r) for r in rows if "adam_kernel" in nk(r))
print("steps profiled", steps)
print("kernel ms per step %.3f, launches per step %.1f" % (sum(tot(r) for r in rows) / steps, sum(calls(r) for r in rows) / steps))
b = json.loads(open("$OUT/${TAG}_b32_bench.json").read().strip().splitlines()[-1])
print("bench ms_per_step", b["ms_per_step"], "value", b["value"])
for r in sorted(rows, key=lambda r: -tot(r))[:45]:
    print("%-100s %6d %8.1f us" % (nk(r)[:100], calls(r), tot(r) * 1e3 / calls(r)))
PY

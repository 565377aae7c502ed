#!/bin/bash
# quick per-kernel averages of one bench run (on the GPU box): tools/quick_trace.sh <tag> [env assignments...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/quick/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" NRL_PROFILE_STEPS=23 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o t -- python $R/bench.py --steps 10 --warmup 3 --no-extras > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/13 steps', tot / 1e6)
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):4d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
PY

#!/bin/bash
# per-kernel averages of 25 NRMS train steps at batch size $1 (default 32) on the GPU box
B=${1:-32}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/quick/b$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o t -- python $R/tools/profile_b32.py $B > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot / 1e6 / 25)
for r in rows[:26]:
    print(f"{r['Name'][:95]:95s} {int(r['Calls'])/25:5.1f}x {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):6.2f}%")
PY

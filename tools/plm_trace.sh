#!/bin/bash
# per-kernel averages of the config-4 (PLM) train step on the GPU box: tools/plm_trace.sh [env assignments...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/quick/plm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -f csv -d $OUT -o t -- python $R/tools/plm_step_time.py --batch 8 --steps 3 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
python - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in rows[:26]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/tot*100:6.2f}%")
PY

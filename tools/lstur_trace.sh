#!/bin/bash
# per-kernel averages of the LSTUR (configs[5]) train step on the GPU box: tools/lstur_trace.sh [env assignments...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/quick/lstur
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -f csv -d $OUT -o t -- python -c "
import sys; sys.path.insert(0, '$R')
import torch, bench
print(bench.extra_lstur(torch.device('cuda:0'), steps=10))
" > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
PY

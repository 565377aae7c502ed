#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 per-kernel statistics of the sibling-model train steps.
#   tools/profile_models.sh "cen mins naml lstur"
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in ${1:-cen mins naml}; do
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/r01_${m}_trace -o r01_${m} -- \
    python $R/tools/lstur_step_time.py --model $m --steps 7 --warmup 3 > $OUT/r01_${m}_trace.log 2>&1
  f=$(find $OUT/r01_${m}_trace -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/r01_${m}_x3_kernel_stats.csv
  tail -1 $OUT/r01_${m}_trace.log
done

#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): collects the rocprofv3 evidence bench.py's roofline line cites.
#   tools/profile_round.sh r03_x3 bf16x3 $(git rev-parse --short=12 HEAD)   (the head is substituted on the calling side: no .git on the box)
# Writes gpurun_out/profiles/<tag>_*; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
ENGINE=${2:-bf16x3}
HEAD=${3:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 10 --warmup 3 --no-extras --engine $ENGINE"
export NRL_PROFILE_STEPS=23   # 3 warm-up + 10 timed + 10 instrumented (bench.py second pass)
# 1. per-kernel time (kernel trace + stats only)
rocprofv3 --kernel-trace --stats -f csv -d $OUT/${TAG}_trace -o $TAG -- $BENCH > $OUT/${TAG}_trace.log 2>&1
# 2./3. HBM traffic counters, each in its own pass (TCC slot limits), no other trace domains
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/${TAG}_fetch -o $TAG -- $BENCH > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/${TAG}_write -o $TAG -- $BENCH > $OUT/${TAG}_write.log 2>&1
# 4. MFMA utilisation counters
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/${TAG}_mfma -o $TAG -- $BENCH > $OUT/${TAG}_mfma.log 2>&1
python $R/tools/summarize_profiles.py $OUT $TAG $ENGINE $HEAD

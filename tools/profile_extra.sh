#!/bin/bash
# Runs ON THE GPU BOX: HBM-traffic and MFMA counters (separate --pmc passes, kernel trace only) for any command, summarised per kernel by
# tools/summarize_profiles.py into gpurun_out/profiles/<tag>_pmc_summary.json.
#   tools/profile_extra.sh r04_plm  python tools/plm_step_time.py --batch 8 --steps 3
#   tools/profile_extra.sh r04_lstur python tools/lstur_step_time.py --steps 10 --warmup 3
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
CMD="$*"
CMD=${CMD//tools\//$R/tools/}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/${TAG}_fetch -o $TAG -- $CMD > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/${TAG}_write -o $TAG -- $CMD > $OUT/${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/${TAG}_mfma -o $TAG -- $CMD > $OUT/${TAG}_mfma.log 2>&1
python $R/tools/summarize_profiles.py $OUT $TAG bf16x3 > /dev/null
python - "$OUT/${TAG}_pmc_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
rows = [(k, e) for k, e in d.items() if k != "_step" and "hbm_bytes_per_launch" in e]
rows.sort(key=lambda ke: -ke[1]["hbm_bytes_per_launch"] * ke[1].get("launches_sampled", 1))
print(f"{'kernel':95s} {'launches':>8s} {'GB/launch':>10s} {'MFMA-busy':>9s}")
for k, e in rows[:22]:
    print(f"{k[:95]:95s} {e.get('launches_sampled', 0):8d} {e['hbm_bytes_per_launch'] / 1e9:10.3f} {e.get('mfma_busy_frac', float('nan')):9.3f}")
PY

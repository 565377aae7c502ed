#!/bin/bash
# tools/ab_env.sh VAR v1 v2 ... : bench.py --no-extras under VAR=v, twice each, interleaved (same box)
VAR=$1; shift
for rep in 1 2; do for v in "$@"; do echo -n "$VAR=$v  "; env $VAR=$v python bench.py --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_ms_per_step'])"; done; done

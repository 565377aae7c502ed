#!/bin/bash
# tools/ab_lstur.sh VAR v1 v2 ...: LSTUR (configs[5]) step time under VAR=v
VAR=$1; shift
for rep in 1 2; do for v in "$@"; do echo -n "$VAR=$v  "; env $VAR=$v python -c "
import torch, bench
print(bench.extra_lstur(torch.device('cuda:0'), steps=15))
" 2>&1 | tail -1; done; done

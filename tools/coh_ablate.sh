#!/bin/bash
# On the GPU box: the coherent kernels with parts switched off (cs_debug_coherent_tuning bits: 1 no scatter-reduce,
# 2 no window flush, 4 no node rows) -- what each part costs.  Results are WRONG with bits set.
for b in 0 1 2 3 4 7; do
  echo "== ablation bits $b"
  CS_SORT=${SORT:-8} CS_ORDER=coherent CS_ABLATE=$b python tools/stage_time.py 10 2>&1 | grep -E "^backward  |backward_backward|bbb_fused"
done

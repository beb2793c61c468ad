#!/bin/bash
# On the GPU box: the coherent kernels with parts switched off (cs_debug_coherent_tuning bits: 1 no scatter-reduce,
# 2 no window flush, 4 no products / outputs; + 16 * waves per workgroup) -- what each part costs.  Results are WRONG with
# bits set.
for b in ${BITS:-0 1 2 4 7}; do
  echo "== ablation bits $b  waves/wg ${WPB:-2}"
  CS_SORT=${SORT:-8} CS_ORDER=coherent CS_CHUNK=${CK:-0} CS_ABLATE=$((b + 16 * ${WPB:-2})) python tools/stage_time.py 10 2>&1 | grep -E "^backward  |backward_backward|bbb_fused"
done

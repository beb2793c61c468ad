#!/bin/bash
# On the GPU box: the coherent kernels at the headline shapes over chunk sizes (samples per wave) and waves per workgroup.
for ck in ${CKS:-512 1024 2048}; do
  for wpb in ${WPBS:-1 2 4}; do
    echo "== chunk $ck waves/wg $wpb"
    CS_SORT=8 CS_ORDER=coherent CS_CHUNK=$ck CS_ABLATE=$((16 * wpb)) python tools/stage_time.py 10 2>&1 | grep -E "^forward|^backward  |backward_backward|bbb_fused"
  done
done

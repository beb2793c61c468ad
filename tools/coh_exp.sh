#!/bin/bash
# On the GPU box: quick timing of the coherent kernels (ordered points, headline shapes), whole and with parts off.
for ck in ${CKS:-512 1024}; do
 for b in ${BITS:-0 7}; do
  echo "== chunk $ck bits $b"
  CS_SORT=8 CS_ORDER=coherent CS_CHUNK=$ck CS_ABLATE=$((b + 16 * ${WPB:-1})) python tools/stage_time.py 10 2>&1 | grep -E "^forward|^backward  |backward_backward|bbb_fused"
 done
done

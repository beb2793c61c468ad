#!/bin/bash
# On the GPU box: the ordered-points step over samples-per-wave settings of the four coherent stages (batches of 64), no rebuild.
export COSINESAMPLER_DEBUG=1
R=$(cd "$(dirname "$0")/.." && pwd)
for ck in ${CKS:-"6,8,6,6 4,8,6,6 3,8,6,6 2,8,6,6 4,8,4,4 4,6,6,6 4,8,8,8 4,4,4,4"}; do
  CS_CHUNKS=$ck python $R/tools/ordered_step.py ${STEPS:-20} 2>&1 | grep chunks
done

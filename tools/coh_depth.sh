#!/bin/bash
# On the GPU box: the coherent kernels (ordered points, headline shapes) for the default library and alt_<name>.so variants
# built with tools/ab.sh (UNIT=cs_coherent), e.g. other prefetch depths.
R=$(cd "$(dirname "$0")/.." && pwd)
for v in default "$@"; do
  if [ $v = default ]; then unset COSINESAMPLER_LIB; else export COSINESAMPLER_LIB=$R/cosinesampler_amd/lib/alt_$v.so; fi
  echo "== library $v"
  CS_SORT=8 CS_ORDER=coherent python $R/tools/stage_time.py 10 2>&1 | grep -E "^forward|^backward|bbb_fused"
done

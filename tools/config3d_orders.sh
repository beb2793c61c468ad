#!/bin/bash
# On the GPU box: BASELINE configs[3] on drawn / ordered points, plan rebuilt / cached, for the default library and variants.
R=$(cd "$(dirname "$0")/.." && pwd)
for v in default "$@"; do
  if [ $v = default ]; then unset COSINESAMPLER_LIB; else export COSINESAMPLER_LIB=$R/cosinesampler_amd/lib/alt_$v.so; fi
  echo "== library $v"; python $R/tools/config3d_orders.py 2>&1 | grep "3D config"
done

#!/bin/bash
# On the GPU box: the summing kernels (tools/sumn_time.py) for the default library and alt_<name>.so variants (tools/ab.sh build,
# UNIT=cs_coherent_sum).
R=$(cd "$(dirname "$0")/.." && pwd)
for v in default "$@"; do
  if [ $v = default ]; then unset COSINESAMPLER_LIB; else export COSINESAMPLER_LIB=$R/cosinesampler_amd/lib/alt_$v.so; fi
  echo "== library $v"; python $R/tools/sumn_time.py 2>&1 | grep -E "sum_n"
done

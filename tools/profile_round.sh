#!/bin/bash
# Run on the GPU box (gpurun): kernel trace + stats and PMC passes of the default bench, summaries under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-round1}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-helmholtz > $R/gpurun_out/prof_$TAG.log 2>&1 || echo "kernel-trace failed"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${TAG}_$tag -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-helmholtz > $R/gpurun_out/pmc_${TAG}_$tag.log 2>&1 || echo "pmc $tag failed"
done

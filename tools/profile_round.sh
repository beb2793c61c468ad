#!/bin/bash
# Run on the GPU box (gpurun): kernel trace + stats and the PMC passes of the default bench (2D headline, the other
# shapes and the autograd-driven Helmholtz step run in the same process), summaries under gpurun_out/.
#   bash tools/profile_round.sh round3      then, in the repo:  python tools/pmc_to_traffic.py round3
# One rocprofv3 run per counter group (the TCC block has 4 slots; FETCH_SIZE costs 3, WRITE_SIZE 2), --pmc never combined
# with tracing.  Each run is bounded: a profiler that hangs must not take the box with it.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-round4}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1 || echo "kernel-trace failed"
echo "kernel trace done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_${TAG}_$i
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "pmc group $i ($set) failed"
  echo "pmc group $i done"
done
# the autograd-driven Helmholtz step alone (1 warm-up + 3 steps): kernel stats and the two byte counters over every kernel
rm -rf $R/gpurun_out/prof_${TAG}_helm
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_helm -- python $R/tools/helmholtz_profile.py > $R/gpurun_out/prof_${TAG}_helm.log 2>&1 || echo "helmholtz kernel-trace failed"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${TAG}_helm_$c
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${TAG}_helm_$c -- python $R/tools/helmholtz_profile.py > $R/gpurun_out/pmc_${TAG}_helm_$c.log 2>&1 || echo "helmholtz pmc $c failed"
  echo "helmholtz pmc $c done"
done

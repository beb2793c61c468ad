#!/bin/bash
# On the GPU box: per-kernel totals of the autograd-driven Helmholtz step (tools/helmholtz_profile.py: 1 warm-up + 3 timed
# steps) under rocprofv3 --kernel-trace --stats; prints sampler kernels vs everything else (torch glue).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_helm
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_helm -- python $R/tools/helmholtz_profile.py > $R/gpurun_out/prof_helm.log 2>&1 || { echo failed; tail -5 $R/gpurun_out/prof_helm.log; exit 1; }
tail -1 $R/gpurun_out/prof_helm.log
python - $(ls $R/gpurun_out/prof_helm/*/*kernel_stats.csv | head -1) <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ours=[r for r in rows if "cs::" in r["Name"] or "zero_fill" in r["Name"] or "_ZN2cs" in r["Name"]]
other=[r for r in rows if r not in ours]
steps=6.0 if __import__('os').environ.get('HELM') in ('sorted', 'summed') else 4.0   # (ordered points: 3 warm-up steps)
t=lambda rs: sum(float(r["TotalDurationNs"]) for r in rs)/1e6/steps
print("sampler kernels %.2f ms/step, torch glue %.2f ms/step" % (t(ours), t(other)))
for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:40]:
    print("  %-70s calls/step %5.1f avg %8.1f us  %6.2f ms/step" % (r["Name"].split("(")[0][-70:], int(r["Calls"])/steps, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/steps))
PY

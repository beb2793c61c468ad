#!/bin/bash
# On the GPU box: kernel trace + stats of the four stages with ordered points (tools/stage_time.py), gpurun_out/cohtrace*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/cohtrace
CS_SORT=8 CS_ORDER=coherent CS_CHUNK=${CK:-512} CS_ABLATE=${ABL:-16} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cohtrace -- python $R/tools/stage_time.py 10 > $R/gpurun_out/cohtrace.log 2>&1 || echo "trace failed"
python - $R/gpurun_out/cohtrace <<'PY' | tee $R/gpurun_out/cohtrace_stats.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-110s calls %5s avg %9.1f us  total %6.2f%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY

#!/bin/bash
# On the GPU box: per-kernel averages of tools/pipe.py for each force mode given (default "2 4"), under rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in ${@:-2 4}; do
  rm -rf $R/gpurun_out/pp_$m
  CS_FORCE=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp_$m -- python $R/tools/${PIPE:-pipe.py} > $R/gpurun_out/pp_$m.log 2>&1 || { echo "mode $m failed"; tail -5 $R/gpurun_out/pp_$m.log; exit 1; }
  echo "== force $m"
  python - $(ls $R/gpurun_out/pp_$m/*/*kernel_stats.csv | head -1) <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "cs::" in n or "zero_fill" in n or "_ZN2cs" in n:
        print("  %-64s calls %3s avg %8.1f us" % (n.split("(")[0][-64:], r["Calls"], float(r["AverageNs"])/1e3))
PY
done

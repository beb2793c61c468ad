#!/bin/bash
# On the GPU box: the headline step at P = 2^20 and at P a little off the power of two (are the (N,C,P) streams' rows, a power of
# two apart, meeting in the same HBM channels?) -> gpurun_out/r4_pitch.txt
R=$GRAFT_REPO_ROOT
for P in ${PS:-1048576 1052672 1056768 1114112}; do
  python $R/bench.py --no-cpu-baseline --no-helmholtz --steps 20 --points $P > /tmp/b_$P.json 2>/dev/null
  python - $P /tmp/b_$P.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
P = int(sys.argv[1])
o = j.get("presorted_points")
print("P=%d  drawn %.3f ms  %.1f Msamples/s | " % (P, j["ms_per_step"], j["value"])
      + " ".join("%s %.3f" % (k[:8], v) for k, v in j["stages_ms"].items())
      + (" || ordered %.3f ms %.1f Ms/s" % (o["ms_per_step"], o["Msamples_per_s"]) if o else ""), flush=True)
PY
done

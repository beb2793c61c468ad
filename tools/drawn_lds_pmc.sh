#!/bin/bash
# On the GPU box: LDS counters of the drawn-points kernels of the headline step (bench.py --no-helmholtz, 2 steps).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/drawnpmc
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/drawnpmc -- python $R/bench.py --no-helmholtz --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/drawnpmc.log 2>&1 || echo failed
python - $R/gpurun_out/drawnpmc <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "cs::tiled" not in k and "pack_cl4" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot, key=lambda k: -tot[k]["SQ_WAVE_CYCLES"]):
    c = tot[k]; w = max(c["SQ_WAVES"], 1)
    print("%-60s waves %9.0f  per wave: LDS insts %7.1f idx_active %8.1f conflict %8.1f (%.0f %%) wave_cycles %9.0f" % (
        k[:60], w, c["SQ_INSTS_LDS"] / w, c["SQ_LDS_IDX_ACTIVE"] / w, c["SQ_LDS_BANK_CONFLICT"] / w,
        100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_WAVE_CYCLES"] / w))
PY

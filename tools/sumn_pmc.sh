#!/bin/bash
# On the GPU box: LDS counters of the summing kernels (tools/sumn_time.py) with parts switched off (CS_ABLATE bits as
# tools/coh_ablate.sh): where do the bank conflicts come from?  One rocprofv3 --pmc pass per setting.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in ${BITS:-0 1 4 5 2048}; do
  rm -rf $R/gpurun_out/sumnpmc_$b
  CS_ABLATE=$b timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/sumnpmc_$b -- python $R/tools/sumn_time.py > $R/gpurun_out/sumnpmc_$b.log 2>&1 || echo "pass $b failed: $(tail -2 $R/gpurun_out/sumnpmc_$b.log)"
done
python - $R/gpurun_out <<'PY'
import csv, glob, sys, collections
for d in sorted(glob.glob(sys.argv[1] + "/sumnpmc_*/")):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "coh::stage" not in k or "true, true>" not in k: continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    print("==", d.rstrip("/").split("_")[-1])
    for k in sorted(tot):
        c = tot[k]; w = c["SQ_WAVES"]
        print("  %-62s per wave: LDS insts %7.0f  idx_active %8.0f  bank_conflict %8.0f  wave_cycles %9.0f" % (
            k.replace("void cs::coh::", ""), c["SQ_INSTS_LDS"] / w, c["SQ_LDS_IDX_ACTIVE"] / w, c["SQ_LDS_BANK_CONFLICT"] / w, c["SQ_WAVE_CYCLES"] / w))
PY

#!/bin/bash
# On the GPU box: SQ counters of the coherent kernels at the headline shapes with ordered points, one rocprofv3 --pmc pass per
# group of at most 8 SQ counters (MI355X_MICROARCH.md, rocprofv3 PMC slots), summed per kernel into gpurun_out/coh_pmc.txt.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
  "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  rm -rf $R/gpurun_out/cohpmc_$i
  CS_SORT=${SORTED-8} CS_ORDER=${ORDER:-coherent} CS_CHUNK=${CK:-512} CS_ABLATE=${ABL:-16} timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/cohpmc_$i -- python $R/tools/stage_time.py 2 > $R/gpurun_out/cohpmc_$i.log 2>&1 || echo "pmc group $i failed: $(tail -2 $R/gpurun_out/cohpmc_$i.log)"
done
python - $R/gpurun_out <<'PY' | tee $R/gpurun_out/${OUT:-coh_pmc}.txt
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/cohpmc_*/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "cs::" not in k or "zero_fill" in k or "plan_" in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], f) not in seen and r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS"):
            seen.add((r["Dispatch_Id"], f)); calls[(k, f)] += 1
for k in sorted(tot):
    n = max(v for (kk, f), v in calls.items() if kk == k)
    print("==", k, "dispatches", n)
    w = tot[k].get("SQ_WAVES", 0) / n
    for c in sorted(tot[k]):
        v = tot[k][c] / n
        print("   %-28s %14.0f   per wave %10.1f" % (c, v, v / w if w else 0))
PY

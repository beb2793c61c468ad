#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of tools/l2_pmc.sh (gpurun_out/l2pmc_<i>/) into one table:
kernel of tools/microbench_l2.hip x counter, plus the kernel's duration in the same (counter-collecting) run.
    python tools/l2_pmc_fold.py [out.json]
"""
import collections, csv, glob, json, os, sys

out = sys.argv[1] if len(sys.argv) > 1 else "profiles/round2_sq_tcp_summary.json"
rows = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/l2pmc_*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("__amd") or k in ("make_perm", "fill_grid"):
            continue
        d = rows.setdefault(k, collections.OrderedDict())
        d[r["Counter_Name"]] = float(r["Counter_Value"])     # one launch per kernel (`microbench_l2 once`)
        d.setdefault("_ns_in_pmc_run", []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in rows.items():
    ns = d.pop("_ns_in_pmc_run")
    d["ns_in_pmc_run_min"] = min(ns)
json.dump({"source": "tools/l2_pmc.sh: rocprofv3 --pmc, one pass per group, tools/microbench_l2 once; raw counter values "
                     "summed over the 8 XCDs x 16 channels (TCC) / 256 CUs (TCP)", "kernels": rows},
          open(out, "w"), indent=1)
names = []
for d in rows.values():
    for c in d:
        if c not in names:
            names.append(c)
for c in names:
    print("%-40s" % c + " ".join("%12.4g" % rows[k].get(c, float("nan")) for k in rows))
print("%-40s" % "" + " ".join("%12s" % k[:12] for k in rows))

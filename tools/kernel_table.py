#!/usr/bin/env python3
"""profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/profile_round.sh) -> profiles/<tag>_kernel_table.txt:
the sampler's own kernels (cs::*), calls and average / min / max duration in microseconds, longest total first, with the
step times the profiled process printed itself.      python tools/kernel_table.py [tag]"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "round4"


def newest(pattern):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", pattern))
    return max(files, key=os.path.getmtime) if files else None


for sub, name in (("prof_%s" % tag, "%s_kernel_stats.csv" % tag), ("prof_%s_helm" % tag, "%s_helmholtz_kernel_stats.csv" % tag)):
    src = newest(os.path.join(sub, "*", "*_kernel_stats.csv"))
    if src:
        shutil.copyfile(src, os.path.join(ROOT, "profiles", name))
rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag))))
line = None
log = os.path.join(ROOT, "gpurun_out", "prof_%s.log" % tag)
if os.path.exists(log):
    for l in open(log):
        if l.startswith("{") and '"metric"' in l:
            line = json.loads(l)
out = ["rocprofv3 --kernel-trace --stats of `python bench.py --steps 5 --warmup 2 --no-cpu-baseline` (tools/profile_round.sh %s;" % tag,
       "full table: %s_kernel_stats.csv; this file: tools/kernel_table.py).  Under the profiler the clocks run lower than in a" % tag,
       "plain run (profiles/%s_bench.json)." % tag]
if line:
    out.append("The profiled process itself printed: %.2f ms per drawn-points step, %.2f ordered, %.2f for configs[3]."
               % (line["ms_per_step"], line["presorted_points"]["ms_per_step"], line["config_3d"]["ms_per_step"]))
out += ["", "calls   avg us   min us   max us   kernel"]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    k = r["Name"].split("(")[0].replace("void ", "")
    if "cs::" not in k:
        continue
    out.append("%5d %8.1f %8.1f %8.1f   %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                              float(r["MaxNs"]) / 1e3, k))
open(os.path.join(ROOT, "profiles", "%s_kernel_table.txt" % tag), "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))

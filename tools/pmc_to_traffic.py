#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE collections (gpurun_out/pmc_<tag>_*/) into
profiles/<tag>_pmc_summary.json and profiles/stage_traffic.json (HBM-side bytes per launch of each
bench stage = sum over the kernels that stage launches).  Counters are reported raw:
bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, no gfx950 read-side correction applied (our stream reads are
4 B/lane and the gathers are 16 B/lane from scattered lines -- neither is the calibrated 16 B/lane
coalesced pattern of MI355X_MICROARCH.md, whose FETCH_SIZE reads exactly half the bytes; the true
read traffic therefore lies between the raw figure and twice it)."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
    files = sorted(glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (tag, c)), key=os.path.getmtime)
    for f in files[-1:]:                      # the newest collection only
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: {"launches": len(x), "per_launch": x[-1]} for c, x in v.items()} for k, v in agg.items() if "cs::" in k}
json.dump(summ, open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1, sort_keys=True)
def kb(kname):
    for k, v in summ.items():
        if k.startswith(kname):
            return (v.get("FETCH_SIZE", {}).get("per_launch", 0) + v.get("WRITE_SIZE", {}).get("per_launch", 0)) * 1024
    return 0
T = 16 * 16 * 256 * 256 * 4
plan = sum(kb("cs::tiled::" + k) for k in ("plan_count", "plan_scan_chunks", "plan_scan_tiles", "plan_scatter", "plan_tile_sort"))
stage = {
    "forward": kb("cs::tiled::pack_channels_last") + kb("cs::tiled::point_forward"),
    "backward": plan + kb("cs::tiled::point_backward<") + kb("cs::tiled::tile_scatter<4, false>") + T,
    "backward_backward": kb("cs::tiled::point_bb<") + kb("cs::tiled::tile_scatter<4, false>") + T,
    "bbb_fused": kb("cs::tiled::point_bbb<") + kb("cs::tiled::tile_scatter<4, true>") + T,
}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, python bench.py --steps 1 --warmup 1 (tools/profile_round.sh %s); raw, see tools/pmc_to_traffic.py" % tag,
           "bytes_per_launch": stage}, open("profiles/stage_traffic.json", "w"), indent=1)
print(json.dumps(stage, indent=1))

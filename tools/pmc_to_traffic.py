#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of tools/profile_round.sh (gpurun_out/pmc_<tag>_<i>/) into
profiles/<tag>_pmc_summary.json (per kernel) and profiles/stage_traffic.json (HBM-side bytes per launch of each bench stage
= sum over the kernels that stage launches), stamped with the digest of the device sources so that bench.py only quotes
it for the kernels it was measured on.

Three figures per kernel, all per launch:
  raw        (FETCH_SIZE + WRITE_SIZE) * 1024                     what rocprofv3 prints
  corrected  (2 * FETCH_SIZE + WRITE_SIZE) * 1024                 /opt/skills/guides/MI355X_MICROARCH.md section HBM: on gfx950
             FETCH_SIZE tallies the 128-byte read requests at 64 bytes; WRITE_SIZE is exact.  THIS is `roofline.traffic`.
  by_size    32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B + 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B)
             the fabric requests counted by their own size counters: an independent check of the correction (all our
             kernels' reads turn out to be 128-byte requests, streams, gathers and row fetches alike).
    python tools/pmc_to_traffic.py [tag]
"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "round4"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
def newest(pattern):
    """one CSV per pass directory: gpurun merges a new run's files next to those of earlier runs"""
    by_dir = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(os.path.dirname(f))
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return [by_dir[d] for d in sorted(by_dir)]


for f in newest(os.path.join(ROOT, "gpurun_out", "pmc_%s_[0-9]*" % tag, "*", "*counter_collection.csv")):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "cs::" in k or "zero_fill" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {}
for k, v in agg.items():
    # bench.py --steps 1 --warmup 1: a kernel's launch 0 is the warm-up step, launch 1 the timed step of the FIRST section
    # that uses it -- the headline step (drawn points, then ordered points), then the other shapes, then the autograd-driven
    # Helmholtz steps, whose launches of the same kernels see other cotangent layouts (and, after a change of the points'
    # order, one or two calls on the wrong path).  So: launch 1, not the last one.
    c = {name: vals[1] if len(vals) > 1 else vals[-1] for name, vals in v.items()}
    d = {"launches_seen": max(len(x) for x in v.values()), "counters": c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        d["corrected"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    if "TCC_EA0_RDREQ_128B_sum" in c and "TCC_EA0_WRREQ_sum" in c:
        d["by_size"] = (32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0)
                        + 128 * c["TCC_EA0_RDREQ_128B_sum"] + 64 * c.get("TCC_EA0_WRREQ_64B_sum", 0)
                        + 32 * (c["TCC_EA0_WRREQ_sum"] - c.get("TCC_EA0_WRREQ_64B_sum", 0)))
    summ[k] = d
json.dump(summ, open(os.path.join(ROOT, "profiles", "%s_pmc_summary.json" % tag), "w"), indent=1, sort_keys=True)


def total(prefixes, field):
    t = 0.0
    for pre in prefixes:
        # "name$": that kernel only; otherwise every kernel whose name starts so (the plan's scan kernels, template arguments)
        hit = [v for k, v in summ.items() if (k == pre[:-1] if pre.endswith("$") else k.startswith(pre))]
        if not hit:
            print("warning: no kernel named", pre, file=sys.stderr)
        for v in hit:
            t += v.get(field, 0.0)
    return t


T = 16 * 16 * 256 * 256 * 4      # the clear of the step's accumulator (round 4: ONE per step, torch's fill kernel, charged
                                 # to the first scatter stage; the one conversion of the ordered step to the last)
plan = ["cs::tiled::plan_count", "cs::tiled::plan_scan_chunks", "cs::tiled::plan_scan_tiles", "cs::tiled::plan_scatter",
        "cs::tiled::plan_tile_sort"]
stages = {
    "forward": ["cs::pack_cl4$", "cs::tiled::point_forward<0, 4, float>"],
    "plan": plan,   # built once per step, used by the three backward stages (bench.py times it as its own span)
    "backward": ["cs::tiled::point_backward<0, 4, true, float>", "cs::tiled::tile_scatter<4, 0, true>"],
    "backward_backward": ["cs::tiled::point_bb<0, 4, false, 2, float>", "cs::tiled::tile_scatter<4, 2, false>"],
    "bbb_fused": ["cs::tiled::point_bbb<0, 4, true, true, float>", "cs::tiled::tile_scatter<4, 3, false>"],
    # the headline step on ORDERED points (bench.py presorted_points): coherent kernels, no plan; the three scatter stages add
    # into one channels-last accumulator: cleared once (T bytes written), unpacked once (cs::unpack_cl4) after the last stage
    "ordered_forward": ["cs::pack_cl4$", "cs::coh::stage<0, 4, 0, false, true, float, true, false>$"],
    "ordered_backward": ["cs::coh::stage<0, 4, 1, false, true, float, true, false>$"],
    "ordered_backward_backward": ["cs::coh::stage<0, 4, 2, false, true, float, true, false>$"],
    "ordered_bbb_fused": ["cs::coh::stage<0, 4, 3, true, true, float, true, false>$", "cs::unpack_cl4"],
    # BASELINE configs[3], same process: 3D smooth-step N=8 C=8 128^3 P=2^19 (accumulator clear not included)
    "3d_forward": ["cs::pack_cl4_zcol", "cs::cl::forward<3, 2, 2, float>"],
    "3d_plan": ["cs::tiles3::plan_count3t", "cs::tiled::plan_scan_chunks", "cs::tiled::plan_scan_tiles", "cs::tiles3::plan_scatter3t"],
    "3d_backward": ["cs::cl::backward<3, 2, 2, 2, float>", "cs::tiles3::tile3_scatter<2, 0>"],
    "3d_backward_backward": ["cs::cl::backward_backward<3, 2, 2, false, 2, float>", "cs::tiles3::tile3_scatter<2, 1>"],
    "3d_bbb_fused": ["cs::cl::bbb<3, 2, 2, 2, float>", "cs::tiles3::tile3_scatter<2, 2>"],
}
import bench
out = {"source": "rocprofv3 --pmc, one pass per counter group, python bench.py --steps 1 --warmup 1 (tools/profile_round.sh %s); "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950 correction of MI355X_MICROARCH.md (128-byte read "
                 "requests are tallied at 64 bytes); raw and by-request-size figures alongside" % tag,
       "csrc_digest": bench.csrc_digest(),
       "bytes_per_launch": {}, "raw": {}, "by_request_size": {}}
for st, ks in stages.items():
    extra = T if st in ("backward", "ordered_backward") else 0
    out["bytes_per_launch"][st] = total(ks, "corrected") + extra
    out["raw"][st] = total(ks, "raw") + extra
    out["by_request_size"][st] = total(ks, "by_size") + extra
# the Helmholtz step (tools/helmholtz_profile.py under --pmc, 4 steps): every kernel of the process, ours and torch's
helm = {"ours": collections.defaultdict(float), "torch": collections.defaultdict(float)}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in newest(os.path.join(ROOT, "gpurun_out", "pmc_%s_helm_%s" % (tag, c), "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            who = "ours" if ("cs::" in k or "zero_fill" in k or "_ZN2cs" in k) else "torch"
            helm[who][r["Counter_Name"]] += float(r["Counter_Value"])
if helm["ours"]:
    HSTEPS = 4.0
    out["pixel_helmholtz_autograd"] = {
        who: {"raw": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / HSTEPS,
              "corrected": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / HSTEPS} for who, v in helm.items()}
    out["pixel_helmholtz_autograd"]["what"] = ("bytes per step of BASELINE configs[2] (N=16 C=16 256^2 P=2^20, reference "
                                               "pattern with grid.repeat), sampler kernels vs torch's own (sum over n, MLP, "
                                               "accumulations), tools/helmholtz_profile.py: 4 steps under --pmc")
json.dump(out, open(os.path.join(ROOT, "profiles", "stage_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

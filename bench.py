#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sampler hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Msamples/s of forward + 3x backward, 2D cosine, multicell,
N=16 C=16 H=W=256, P=2^20 points PER GPU (configs[1]; configs[4] = the same per-GPU work on
8 GPUs, P=2^23 in total => weak scaling).  A *sample* is one (n,p) output location, all C channels.

One step = one pass of the whole hot path over one batch of synthetic, HBM-resident inputs:
    forward (K1) -> the point plan of the grid (shared by the backward stages, rebuilt every step) ->
    backward (K2: grad_input + grad_grid) -> backward_backward (K3, gOutInput absent)
    -> fused third backward (K4 + the reference's extra K3).  What the step delivers of the three input-shaped
    gradients is their SUM (what autograd accumulates into cells.grad): the stages add into one step accumulator
    (ops.StepContext(accumulate=True), cs_cotangent_layout.accumulate_grad_input) -- one clear and one layout conversion
    per step -- and for N>1 that sum is all-reduced ONCE over the ranks (RCCL; --reduce once, the default).
    --reduce per_stage keeps the three gradients apart and starts an asynchronous all-reduce for each as soon as its
    stage is enqueued (cosinesampler_amd.dist.GradReducer), the step waiting once at its end.
All calls go through the C ABI (cosinesampler_amd.ops -> libcosine_sampler_hip.so).

Prints ONE JSON line on rank 0.  `roofline` is for the slowest stage kernel(s) (HIP-event time on
the launch stream, algorithmic bytes from BASELINE.md section 3); `pipeline_roofline_frac` is the
same ratio for the whole step.  `cpu_baseline` times this repo's pure-PyTorch composite of the op
(oracle/composite.py, the same kind of path as the reference's test/grid_sampler.py) through
torch.autograd on the host cores, on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md

# what each timed stage launches at this config (names as rocprofv3 prints them, profiles/round2_kernel_stats.csv);
# a stage's HIP-event time is the sum of these plus the 64 MiB clear of its grad_input
STAGE_KERNELS = {
    "forward": ["cs::pack_cl4", "cs::tiled::point_forward<0, 4, float>"],
    "plan": ["cs::tiled::plan_count/scan_chunks/scan_tiles/scatter/tile_sort"],
    "backward": ["cs::tiled::point_backward<0, 4, true, float>", "cs::tiled::tile_scatter<4, 0, true>"],
    "backward_backward": ["cs::tiled::point_bb<0, 4, false, 2, float>", "cs::tiled::tile_scatter<4, 2, false>"],
    "bbb_fused": ["cs::tiled::point_bbb<0, 4, true, true, float>", "cs::tiled::tile_scatter<4, 3, false>"],
}


def csrc_digest():
    """Identity of the kernels the numbers belong to: sha256 over the device sources + the C ABI header."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "cosinesampler_amd", "csrc")
    for name in sorted(os.listdir(base)):
        h.update(name.encode())
        h.update(open(os.path.join(base, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "cosine_sampler.h"), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(S, C, d, T):
    """BASELINE.md section 3 / SURVEY.md 8(d): compulsory traffic, every tensor element once."""
    return {
        "forward": 4 * S * (d + C) + T,
        "backward": 4 * S * (2 * d + C) + 2 * T,
        "backward_backward": 4 * S * (3 * d + 2 * C) + 2 * T,
        "bbb_fused": 4 * S * (3 * d + 3 * C) + 2 * T,
    }


def cpu_baseline(N, C, H, seconds_budget=20.0):
    """The PyTorch-autograd CPU path on a bounded sample of the same workload."""
    from oracle import composite
    # the GPU box hands a 1-GPU job a 16-CPU share of a 256-thread host; asking torch for every
    # hardware thread oversubscribes that share badly (measured: 50x slower)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)

    def run(P):
        g = torch.Generator().manual_seed(0)
        cells = torch.rand(N, C, H, H, generator=g).requires_grad_(True)
        xy = torch.rand(P, 2, generator=g) * 2 - 1
        grid = xy.view(1, 1, P, 2).repeat(N, 1, 1, 1).requires_grad_(True)
        gOut = torch.randn(N, C, 1, P, generator=g).requires_grad_(True)
        cG = torch.randn(N, 1, P, 2, generator=g)
        hG = torch.randn(N, 1, P, 2, generator=g)
        hO = torch.randn(N, C, 1, P, generator=g)
        t0 = time.perf_counter()
        out = composite.grid_sample_nd(cells, grid, "cosine", True, True)
        gI, gG = torch.autograd.grad(out, (cells, grid), gOut, create_graph=True)
        bbI, bbG, bbO = torch.autograd.grad((gG * cG).sum(), (cells, grid, gOut), create_graph=True)
        tI, tO = torch.autograd.grad((bbG * hG).sum() + (bbO * hO).sum(), (cells, gOut))
        return time.perf_counter() - t0

    P = 1 << 12
    t = run(P)                     # warm-up + calibration
    while t < seconds_budget / 4 and P < (1 << 20):   # grow the sample until one pass takes ~5 s or more
        P *= 2
        t = run(P)
    t = min(t, run(P))
    S = N * P
    return {"value": S / t / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "oracle/composite.py via torch.autograd (fwd + 3 grad levels), 2D cosine multicell "
                      "N=%d C=%d H=W=%d P=%d (%d samples) in %.2f s on %d threads" % (N, C, H, P, S, t, cores)}


def helmholtz_step(N, C, H, P, dev, steps=3, broadcast_grid=False, sort_points=False, grid_leaf=False, summed_op=False):
    """BASELINE.json configs[2]: the PIXEL-style Helmholtz step driven entirely by torch.autograd
    (reference test/test_2d.py pattern): u = MLP(sum_n sampler(cells, grid)); u_x, u_y; u_xx, u_yy via the
    second backward; loss = mean((u_xx + u_yy + k^2 u)^2); d loss / d cells via the third backward.
    Returns average milliseconds per step (HIP events), first step excluded."""
    from cosinesampler_amd import CosineSampler2d, CosineSampler2dSum
    g = torch.Generator(device="cpu").manual_seed(7)
    cells = torch.rand(N, C, H, H, generator=g).to(dev).requires_grad_(True)
    W1 = (torch.randn(16, C, generator=g) * 0.5).to(dev)
    W2 = (torch.randn(1, 16, generator=g) * 0.5).to(dev)
    x = (torch.rand(P, 1, generator=g) * 2 - 1).to(dev)
    y = (torch.rand(P, 1, generator=g) * 2 - 1).to(dev)
    if sort_points:      # the collocation set is fixed: order it ONCE by cell (ops.sort_points), as a PIXEL caller would at set-up
        from cosinesampler_amd import ops
        xy, _ = ops.sort_points(torch.cat([x, y], -1), (H, H))
        x, y = xy[:, :1].contiguous(), xy[:, 1:].contiguous()
    x.requires_grad_(True)
    y.requires_grad_(True)
    ones = torch.ones(P, 1, device=dev)

    grid0 = torch.cat([x, y], -1).detach().view(1, 1, P, 2).requires_grad_(True) if grid_leaf else None

    def one_leaf():
        # the same step with the points kept as ONE (1,1,P,2) leaf that every step hands over again: u_x, u_y are the two
        # columns of d u / d grid -- the plan of the (unchanged) grid tensor is found in ops.plan_cache instead of rebuilt
        val = CosineSampler2d.apply(cells, grid0, "zeros", True, "cosine", True)
        u = torch.tanh(val.sum(0).view(C, -1).t() @ W1.t()) @ W2.t()
        (gG,) = torch.autograd.grad(u, grid0, ones, create_graph=True)
        u_x, u_y = gG[0, 0, :, 0:1], gG[0, 0, :, 1:2]
        (hx,) = torch.autograd.grad(u_x, grid0, ones, create_graph=True)
        (hy,) = torch.autograd.grad(u_y, grid0, ones, create_graph=True)
        loss = torch.mean((hx[0, 0, :, 0:1] + hy[0, 0, :, 1:2] + 4.0 * u) ** 2)
        (gc,) = torch.autograd.grad(loss, cells)
        return gc

    def one():
        if grid_leaf:
            return one_leaf()
        grid = torch.cat([x, y], -1).view(1, 1, P, 2)
        if summed_op:                     # sampler(...).sum(0) as ONE op (CosineSampler2dSum, CS_SUM_OVER_N): no (N,C,P) tensor at all
            feat = CosineSampler2dSum.apply(cells, grid, "zeros", True, "cosine", True)[0]
        else:
            if not broadcast_grid:            # the reference pattern (test/test_2d.py:38); broadcast_grid: the same points
                grid = grid.repeat(N, 1, 1, 1)   # handed over once, CS_GRID_BROADCAST (not expressible with the reference op)
            feat = CosineSampler2d.apply(cells, grid, "zeros", True, "cosine", True).sum(0)
        u = torch.tanh(feat.view(C, -1).t() @ W1.t()) @ W2.t()
        u_x, u_y = torch.autograd.grad(u, (x, y), ones, create_graph=True)
        (u_xx,) = torch.autograd.grad(u_x, x, ones, create_graph=True)
        (u_yy,) = torch.autograd.grad(u_y, y, ones, create_graph=True)
        loss = torch.mean((u_xx + u_yy + 4.0 * u) ** 2)
        (gc,) = torch.autograd.grad(loss, cells)
        return gc

    for _ in range(3 if sort_points else 1):     # (the order of the points is measured behind an event: it arrives a call late)
        one()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        one()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def stage_pipeline_ms(dev, dim, N, C, size, P, kernel, steps=10, stream_dtype=None, warm_plan=False, sort_points=False,
                      summed=False):
    """forward + the three backward stages on synthetic inputs of the given shape (any of BASELINE.json's configs or
    the reference test scripts' shapes), fresh StepContext per step -> (ms per step, samples per step).
    summed: the PIXEL pattern as ONE op (CS_SUM_OVER_N: one set of points and cotangents, per-point results summed over
    the tables) -- the same table samples, the same input-shaped gradients, 1/N of the streams."""
    from cosinesampler_amd import multicell_offset, ops
    g = torch.Generator(device="cpu").manual_seed(11)
    cells = torch.rand((N, C) + (size,) * dim, generator=g).to(dev)
    pts = (torch.rand(P, dim, generator=g) * 2 - 1).to(dev)
    if sort_points:          # a fixed point set ordered once by cell (ops.sort_points), as in the 2D headline's presorted_points
        pts, _ = ops.sort_points(pts, (size,) * dim)
    grid = pts.view((1,) * dim + (P, dim)).repeat((1 if summed else N,) + (1,) * (dim + 1)).contiguous()
    oshape = (1 if summed else N, C) + (1,) * (dim - 1) + (P,)
    gOut = torch.randn(oshape, generator=g).to(dev)
    hO = torch.randn(oshape, generator=g).to(dev)
    if stream_dtype is not None:     # the channel-major streams in float16 / bfloat16, read and written natively
        gOut, hO = gOut.to(stream_dtype), hO.to(stream_dtype)
    cG = torch.randn(grid.shape, generator=g).to(dev)
    hG = torch.randn(grid.shape, generator=g).to(dev)
    off = multicell_offset(N, True, dev)

    def one():
        sc = ops.StepContext()
        if summed:
            ops.forward_sum_n(cells, grid, off, 0, True, kernel, True, ctx=sc)
            ops.backward_sum_n(gOut, cells, grid, off, 0, True, True, kernel, True, ctx=sc)
            ops.backward_backward_sum_n(cG, cells, grid, gOut, off, 0, True, kernel, True, ctx=sc)
            ops.bbb_fused_sum_n(cells, grid, gOut, cG, hG, hO, off, 0, True, kernel, True, ctx=sc)
            return
        ops.forward(cells, grid, off, 0, True, kernel, True, ctx=sc, out_dtype=stream_dtype)
        ops.backward(gOut, cells, grid, off, 0, True, True, kernel, True, ctx=sc)
        ops.backward_backward(None, cG, cells, grid, gOut, off, 0, True, False, kernel, True, ctx=sc)
        ops.bbb_fused(cells, grid, gOut, cG, hG, hO, off, 0, True, kernel, True, ctx=sc)

    if warm_plan:        # the grid tensor is the same every step: its plan is kept across steps (ops.plan_cache)
        ops.plan_cache(1)
    try:
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            one()
        e1.record()
        torch.cuda.synchronize()
    finally:
        ops.plan_cache(0)
    return e0.elapsed_time(e1) / steps, N * P


OTHER_SHAPES = [   # (key, what, dim, N, C, size, P, kernel enum)
    ("config_3d", "BASELINE.json configs[3]: 3D smooth-step N=8 C=8 128^3 P=2^19", 3, 8, 8, 128, 1 << 19, 2),
    ("reference_test_shapes", "reference test/test_2d.py shapes: 2D cosine N=96 C=4 16^2 P=100000 (crowded tables)",
     2, 96, 4, 16, 100000, 0),
    ("channels_32", "2D cosine N=16 C=32 256^2 P=2^20: the widest table the fast path takes whole", 2, 16, 32, 256, 1 << 20, 0),
    ("channels_64", "2D cosine N=16 C=64 256^2 P=2^20: two channel ranges of 32 through the fast path (ops: channel groups)",
     2, 16, 64, 256, 1 << 20, 0),
    ("reference_test_shapes_3d", "reference test/test_3d.py shapes: 3D cosine N=50 C=4 16^3 P=100000 (crowded tables)",
     3, 50, 4, 16, 100000, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1 << 20, help="P per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-helmholtz", action="store_true", help="skip the extra autograd-driven PIXEL step timing")
    ap.add_argument("--reduce", choices=["per_stage", "once"], default="once",
                    help="N>1: the stages add into one step accumulator and its sum is all-reduced once per step (1 x 64 MiB, "
                         "SURVEY 8e; default) or every stage's input-shaped gradient is kept apart and all-reduced as soon as "
                         "it is enqueued (3 x 64 MiB, overlapped, then summed)")
    ap.add_argument("--rccl-alone", action="store_true",
                    help="with one rank: still create the RCCL process group and run the gradient all-reduces through it")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    # --rccl-alone: a ONE-rank RCCL group whose collectives really run (tests/test_parity_gpu.py: the communication
    # path of the multi-GPU job, rehearsed on the single GPU a test box has)
    use_dist = world > 1 or args.rccl_alone
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm

    from cosinesampler_amd import multicell_offset, ops
    from cosinesampler_amd.dist import GradReducer

    N, C, H, P, d = 16, 16, 256, args.points, 2
    S = N * P
    T = 4 * N * C * H * H
    pad, align, kern, mc = 0, True, 0, True

    # synthetic inputs, each rank its own chunk of the global point set
    g = torch.Generator(device="cpu").manual_seed(0)
    cells = torch.rand(N, C, H, H, generator=g).to(dev)
    gp = torch.Generator(device="cpu").manual_seed(1000 + rank)
    xy = (torch.rand(P, 2, generator=gp) * 2 - 1).to(dev)
    grid = xy.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()          # PIXEL pattern
    torch.manual_seed(1 + rank)
    gOut = torch.randn(N, C, 1, P, device=dev)
    hO = torch.randn(N, C, 1, P, device=dev)
    cG = torch.randn(N, 1, P, 2, device=dev)
    hG = torch.randn(N, 1, P, 2, device=dev)
    off = multicell_offset(N, mc, dev)
    acc = torch.zeros_like(cells)
    # one step accumulator for the three input-shaped gradients unless the per-stage reduce schedule needs them apart
    accumulate = not (use_dist and args.reduce == "per_stage")

    stage_names = ["forward", "backward", "backward_backward", "bbb_fused"]
    ev = []
    out_keep = []

    def step(record, reduce=True, grid=grid, order=None, ev=ev):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if record else None
        # A fresh StepContext every step: the channels-last copy of `cells` and the point plan of
        # `grid` are rebuilt inside the timed region each step (forward pays the copy, backward the
        # plan), exactly as one CosineSampler2d.apply + its backward chain would.
        sc = ops.StepContext(points_order=order, accumulate=accumulate)
        # every input-shaped gradient starts its sum over the ranks the moment its stage has been enqueued (RCCL's own
        # stream, behind the producing kernels) and the step waits once, at the end: only the last one is exposed
        red = GradReducer(even_alone=args.rccl_alone, enabled=reduce and use_dist, schedule=args.reduce)
        if record:
            e[0].record()
        out = ops.forward(cells, grid, off, pad, align, kern, mc, ctx=sc)
        if record:
            e[1].record()
        # the point plan: a function of the grid alone, used by all three backward stages -- built here so that it is
        # timed as what it is instead of inside whichever stage scatters first (it is part of the step either way)
        if order != "coherent":          # (ordered points: the scatter stages need no plan)
            sc.prepare_plan(cells, grid, off, pad, align, mc)
        if record:
            e[2].record()
        gI, gG = ops.backward(gOut, cells, grid, off, pad, align, True, kern, mc, ctx=sc)
        if not accumulate:
            red.push(gI)
        if record:
            e[3].record()
        bbI, bbG, bbO = ops.backward_backward(None, cG, cells, grid, gOut, off, pad, align, False, kern, mc, ctx=sc)
        if not accumulate:
            red.push(bbI)
        if record:
            e[4].record()
        tI, tO = ops.bbb_fused(cells, grid, gOut, cG, hG, hO, off, pad, align, kern, mc, ctx=sc)
        if accumulate:                   # the sum of the three gradients, in the caller's layout (the third stage's span
            total = red.push(sc.grad_input_sum())          # pays the one conversion), then ONE all-reduce over the ranks
        else:
            red.push(tI)
        if record:
            e[5].record()
            ev.append(e)
        total = red.finish(out=None if accumulate else acc)   # = sum over stages (and ranks) of the input-shaped gradients
        return out, gG, bbG, bbO, tO, total

    def timed(steps, record, reduce=True, **kw):
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(record, reduce, **kw)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(args.warmup):
        step(False)
    # the metric: exactly K steps, nothing but the step in the loop; the per-stage spans come from a second pass whose
    # steps carry six event records each (they cost the pipeline a little: markers between the launches)
    elapsed = timed(args.steps, False)
    timed(args.steps, True)
    no_reduce_ms = None
    if use_dist:   # the same steps without the collectives: what the reduction costs the step
        no_reduce_ms = timed(args.steps, False, reduce=False) / args.steps * 1e3

    spans = {"forward": 0, "plan": 1, "backward": 2, "backward_backward": 3, "bbb_fused": 4}
    stage_ms = {nm: sum(e[i].elapsed_time(e[i + 1]) for e in ev) / len(ev) for nm, i in spans.items()}

    # The same step on the same points in CELL ORDER -- what a PIXEL-style caller hands over after ordering its (fixed)
    # collocation set once with ops.sort_points (reference test/test_2d.py:28-38 draws the set once): the scatter stages
    # then run on the coherent-points kernels (CS_POINTS_COHERENT), no plan.  Reported beside the headline, never instead:
    # `value` stays the step on the points as drawn.
    presorted = None
    if P == (1 << 20) or os.environ.get("CS_BENCH_PRESORTED"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xy_s, perm = ops.sort_points(xy, (H, H), pad, align, mc)
        torch.cuda.synchronize()
        sort_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        xy_s, perm = ops.sort_points(xy, (H, H), pad, align, mc)
        torch.cuda.synchronize()
        sort_ms = min(sort_ms, (time.perf_counter() - t0) * 1e3)
        grid_s = xy_s.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
        ev_s = []
        for _ in range(args.warmup):
            step(False, grid=grid_s, order="coherent")
        el_s = timed(args.steps, False, grid=grid_s, order="coherent")
        timed(args.steps, True, grid=grid_s, order="coherent", ev=ev_s)
        st_s = {nm: sum(e[i].elapsed_time(e[i + 1]) for e in ev_s) / len(ev_s) for nm, i in spans.items()}
        presorted = (el_s / args.steps * 1e3, st_s, sort_ms)
        del grid_s, xy_s, perm
    # The drawn-points step when the grid TENSOR is the same every step (fixed collocation points) and its plan is kept
    # across steps (ops.plan_cache; the sorted grad_output copy is still made anew every step)
    warm_ms = None
    if not use_dist and P == (1 << 20):
        ops.plan_cache(1)
        try:
            for _ in range(2):
                step(False, False)
            warm_ms = timed(args.steps, False, reduce=False) / args.steps * 1e3
        finally:
            ops.plan_cache(0)
    # The same steps captured once into a HIP graph and replayed (torch.cuda.CUDAGraph): nothing in a stage allocates
    # through HIP, synchronises or touches the host, so a caller whose loop is launch-bound can do this; what it removes is
    # the gaps between the ~20 (drawn) / ~14 (ordered) launches of a step.  Beside the headline, which stays eager.
    graph_ms = None
    if not use_dist and P == (1 << 20):
        def replay_ms(**kw):
            step(False, False, **kw)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                keep = step(False, False, **kw)
            for _ in range(2):
                gr.replay()
            torch.cuda.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            for _ in range(args.steps):
                gr.replay()
            g1.record()
            torch.cuda.synchronize()
            ms = g0.elapsed_time(g1) / args.steps
            del gr, keep
            return ms
        try:
            graph_ms = {"ms_per_step": replay_ms()}
            if presorted is not None:
                grid_s = ops.sort_points(xy, (H, H), pad, align, mc)[0].view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
                graph_ms["presorted_ms_per_step"] = replay_ms(grid=grid_s, order="coherent")
                del grid_s
        except RuntimeError as exc:          # capture not possible in this environment: say so, the headline does not depend on it
            graph_ms = {"error": str(exc)[:200]}
        torch.cuda.empty_cache()
    ab = algorithmic_bytes(S, C, d, T)
    dom = max(stage_names, key=lambda k: stage_ms[k])
    achieved = ab[dom] / (stage_ms[dom] * 1e-3)
    ms_per_step = elapsed / args.steps * 1e3
    total_bytes = sum(ab.values())

    # HBM-side bytes of the dominant stage: measured by the PMC passes of tools/profile_round.sh (same command, same
    # config) and folded by tools/pmc_to_traffic.py -- a separate rocprofv3 run, since counters cannot be read in-process.
    # Emitted only when that file was made from THESE kernels (digest of the device sources), else null.
    traffic, traffic_source = None, "not measured for this build (profiles/stage_traffic.json is from other kernels or another config)"
    tpath = os.path.join(ROOT, "profiles", "stage_traffic.json")
    if P == (1 << 20) and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("csrc_digest") == csrc_digest():
                traffic = tj["bytes_per_launch"].get(dom)
                traffic_source = tj.get("source")
        except (ValueError, KeyError):
            pass

    if rank == 0:
        line = {
            "metric": "Msamples/s fwd+3xbwd (2D cosine, C=16, 256^2 grid)",
            "value": world * S * args.steps / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "2D cosine multicell zeros align_corners N=16 C=16 H=W=256 P=%d per GPU: "
                                   "forward + backward + backward_backward + fused third backward%s"
                                   % (P, (" + RCCL all-reduce of the grad_inputs (%s)" % ("3 x 64 MiB, overlapped" if args.reduce == "per_stage" else "1 x 64 MiB, once per step")) if use_dist else ""),
                       "samples_per_step_per_gpu": S, "sharding": "points (P) across ranks"},
            "roofline": {"bound": "hbm", "kernel": dom, "kernels": STAGE_KERNELS[dom], "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": ab[dom], "ms_per_launch": stage_ms[dom]},
            "pipeline_roofline_frac": total_bytes / (ms_per_step * 1e-3) / HBM_PEAK,
            "stages_ms": stage_ms,
            "stages_frac": {k: ab[k] / (stage_ms[k] * 1e-3) / HBM_PEAK for k in stage_names},
        }
        if presorted is not None:
            ms_s, st_s, sort_ms = presorted
            line["presorted_points"] = {
                "ms_per_step": ms_s, "Msamples_per_s": world * S / ms_s / 1e3,
                "roofline_frac": total_bytes / (ms_s * 1e-3) / HBM_PEAK, "sort_ms_once": sort_ms,
                "stages_ms": st_s, "stages_frac": {k: ab[k] / (st_s[k] * 1e-3) / HBM_PEAK for k in stage_names},
                "what": "the same step on the same points after ops.sort_points (cs2d_sort_points: by 8-cell tile, then cell; "
                        "once, at set-up) with the CS_POINTS_COHERENT hint: the three scatter stages run on "
                        "cs::coh::backward / bb / bbb (run reduction on chip, no plan: its span is ~0); results are the "
                        "same for any order, only the time differs"}
        if warm_ms is not None:
            line["ms_per_step_warm_plan"] = warm_ms
        if graph_ms is not None:
            graph_ms["what"] = ("the same step (drawn points) and the ordered-points step captured once into a HIP graph "
                                "(torch.cuda.CUDAGraph) and replayed: the launch gaps of the eager step are gone")
            line["hip_graph"] = graph_ms
        if use_dist:
            line["reduce_schedule"] = args.reduce
            line["ms_per_step_no_reduce"] = no_reduce_ms
            line["allreduce_ms"] = ms_per_step - no_reduce_ms
            line["reduce"] = ("3 RCCL all-reduces of 64 MiB per step (one per input-shaped gradient), each started "
                              "asynchronously when its stage is enqueued, one wait at the end of the step"
                              if args.reduce == "per_stage" else
                              "the three input-shaped gradients are summed locally, then ONE RCCL all-reduce of 64 MiB per step")
        if world == 1 and not args.no_helmholtz:
            del out_keep[:]
            del cells, grid, gOut, hO, cG, hG, acc      # make room: the 3D config holds a 512 MiB table
            torch.cuda.empty_cache()
            for key, what, dim_, n_, c_, size_, p_, kern_ in OTHER_SHAPES:   # the same four stages at other shapes
                ms_o, s_o = stage_pipeline_ms(dev, dim_, n_, c_, size_, p_, kern_, steps=5 if c_ > 16 else 10)
                line[key] = {"ms_per_step": ms_o, "Msamples_per_s": s_o / ms_o / 1e3, "what": what}
                if key == "channels_64":
                    line[key]["vs_two_32_channel_steps"] = ms_o / (2 * line["channels_32"]["ms_per_step"])
                if key == "config_3d":     # fixed collocation points: the plan of the (same) grid tensor kept across steps
                    line[key]["ms_per_step_warm_plan"] = stage_pipeline_ms(dev, dim_, n_, c_, size_, p_, kern_, warm_plan=True)[0]
                    # the same kernels on the same points in cell order (ops.sort_points, 3D): gathers and row fetches of
                    # neighbouring samples share lines
                    line[key]["ms_per_step_sorted_points"] = stage_pipeline_ms(dev, dim_, n_, c_, size_, p_, kern_, sort_points=True)[0]
                    line[key]["ms_per_step_sorted_points_warm_plan"] = stage_pipeline_ms(dev, dim_, n_, c_, size_, p_, kern_, warm_plan=True, sort_points=True)[0]
                    # the config's points are the same for every table (PIXEL): as ONE summed op (CS_SUM_OVER_N in 3D,
                    # round 4) -- the same table samples and input-shaped gradients, one set of (1,C,P) streams
                    line[key]["ms_per_step_summed_op"] = stage_pipeline_ms(dev, dim_, n_, c_, size_, p_, kern_, summed=True)[0]
                    line[key]["ms_per_step_summed_op_sorted_points"] = stage_pipeline_ms(dev, dim_, n_, c_, size_, p_, kern_, summed=True, sort_points=True)[0]
            ms_h, s_h = stage_pipeline_ms(dev, 2, N, C, H, P, 0, stream_dtype=torch.bfloat16)
            line["bf16_streams"] = {"ms_per_step": ms_h, "Msamples_per_s": s_h / ms_h / 1e3,
                                    "what": "the headline step with output / grad_output / grad_grad_out / grad_out_ggout "
                                            "in bfloat16 (CS_STREAM_BF16: native 16-bit stream I/O, fp32 arithmetic, "
                                            "table and grid fp32); reported next to the fp32 headline, not instead of it"}
            line["bf16_streams"]["ms_per_step_sorted_points"] = stage_pipeline_ms(dev, 2, N, C, H, P, 0, stream_dtype=torch.bfloat16,
                                                                                  sort_points=True)[0]
            ms = helmholtz_step(N, C, H, P, dev)
            line["pixel_helmholtz_autograd"] = {
                "ms_per_step": ms, "Msamples_per_s": S / ms / 1e3,
                "what": "configs[2]: CosineSampler2d.apply -> sum over n -> MLP -> u_x,u_y -> u_xx,u_yy -> "
                        "d mean((u_xx+u_yy+4u)^2)/d cells, all through torch.autograd (incl. its .contiguous() "
                        "copies of the expanded gradients and the MLP), same N,C,H,W,P"}
            ms_b = helmholtz_step(N, C, H, P, dev, broadcast_grid=True)
            line["pixel_helmholtz_autograd"]["ms_per_step_broadcast_grid"] = ms_b
            line["pixel_helmholtz_autograd"]["ms_per_step_sorted_points"] = helmholtz_step(N, C, H, P, dev, sort_points=True)
            line["pixel_helmholtz_autograd"]["ms_per_step_sorted_points_broadcast_grid"] = helmholtz_step(
                N, C, H, P, dev, broadcast_grid=True, sort_points=True)
            line["pixel_helmholtz_autograd"]["ms_per_step_sorted_points_summed_op"] = helmholtz_step(
                N, C, H, P, dev, sort_points=True, summed_op=True)
            line["pixel_helmholtz_autograd"]["ms_per_step_summed_op"] = helmholtz_step(N, C, H, P, dev, summed_op=True)
            line["pixel_helmholtz_autograd"]["summed_op"] = (
                "the same step with features = CosineSampler2dSum.apply(cells, points) instead of "
                "CosineSampler2d.apply(cells, points.repeat(N,1,1,1)).sum(0): the summing kernels (CS_SUM_OVER_N) run and no (N,C,P) "
                "tensor exists anywhere in the step; points in the order they were drawn are put into cell order inside the op "
                "(one sort of the P points per step, the (1,C,P) cotangents and results carried over with index selections)")
            from cosinesampler_amd import ops as _ops
            _ops.plan_cache(1)
            try:
                line["pixel_helmholtz_autograd"]["ms_per_step_warm_plan"] = helmholtz_step(N, C, H, P, dev, grid_leaf=True)
            finally:
                _ops.plan_cache(0)
            line["pixel_helmholtz_autograd"]["warm_plan"] = (
                "the points kept as one (1,1,P,2) leaf tensor handed over every step (u_x, u_y = the columns of d u / d grid) "
                "with ops.plan_cache(1): the point plan of the unchanged grid tensor is re-used instead of rebuilt")
            line["pixel_helmholtz_autograd"]["sorted_points"] = (
                "the same step with the (fixed) collocation points ordered once by cell with ops.sort_points: the op measures "
                "the order itself (ops.points_order('auto')) and runs the scatter stages on the coherent-points kernels")
            line["pixel_helmholtz_autograd"]["broadcast_grid"] = (
                "the same step with the points handed over once, a (1,1,P,2) grid (CS_GRID_BROADCAST) instead of "
                "grid.repeat(N,1,1,1): no 128 MiB repeat, its backward sums are taken by the op")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, C, H)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

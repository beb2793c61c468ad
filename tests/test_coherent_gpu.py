"""GPU (MI355X): the coherent-points path (CS_POINTS_COHERENT, cosinesampler_amd/csrc/cs_coherent.cuh) and the
point-ordering helpers, through the C ABI, against the CPU oracle.

The hint is about speed only, so every case is checked on points in the order the helper produces, on half-ordered
points (the adversarial middle: long ordered stretches broken by jumps and by stretches of unordered points) and on
unordered points.  Tolerance: helpers.REL_TOL (1e-5 relative per tensor)."""
import numpy as np
import pytest
import torch

from cosinesampler_amd import _lib, multicell_offset, ops
from helpers import assert_close, offsets, rel_err
from oracle import cs_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _points(P, d, seed, spread=1.1):
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(P, d, generator=g) * (2 * spread) - spread
    pts[0] = -1.0
    pts[1] = 1.0
    pts[2] = 0.0
    return pts


def _order(pts, size, how, pad, align, mc, seed):
    """the point set in one of three orders"""
    if how == "random":
        return pts
    srt, perm = ops.sort_points(pts.to(DEV), size, pad, align, mc)
    srt = srt.cpu()
    if how == "sorted":
        return srt
    # half-ordered: blocks of the ordered set in a shuffled block order, every third block shuffled inside as well
    g = torch.Generator().manual_seed(seed)
    P = srt.shape[0]
    nb = 23
    bounds = [P * i // nb for i in range(nb + 1)]
    out = []
    for j, b in enumerate(torch.randperm(nb, generator=g).tolist()):
        blk = srt[bounds[b]:bounds[b + 1]]
        if j % 3 == 0:
            blk = blk[torch.randperm(blk.shape[0], generator=g)]
        out.append(blk)
    return torch.cat(out)


def _key_reference(pts, size, pad, align, mc):
    """numpy restatement of the ordering key (cs_sort.hip cell_key): fp32 source index of table 0, 8-cell tiles, quads, cells"""
    pts = pts.numpy().astype(np.float32)
    d = pts.shape[1]
    sizes = list(size)[::-1]            # x (W) first
    tile = np.zeros(len(pts), dtype=np.int64)
    quad = np.zeros(len(pts), dtype=np.int64)
    sub = np.zeros(len(pts), dtype=np.int64)
    last = np.zeros(len(pts), dtype=bool)
    for j in range(d - 1, -1, -1):
        s = sizes[j]
        g = pts[:, j]
        if align:
            ss = s - 1 if mc else s
            i = ((g + np.float32(1)) * np.float32(0.5)).astype(np.float32) * np.float32(ss - 1)   # fmaf(x, s-1, 0) rounds once
        else:
            i = ((g + np.float32(1)) * np.float32(s) - np.float32(1)).astype(np.float32) * np.float32(0.5)
        assert pad == 0
        u = np.floor(i.astype(np.float32)).astype(np.int64) + 1
        last |= (u < 0) | (u > s)
        nt = s // 8 + 1
        tile = tile * nt + np.clip(u, 0, s) // 8
        quad = quad * 4 + (np.clip(u, 0, s) % 8) // 2      # 2 x 2 (x 2) cells, row-major inside the tile
        sub = sub * 2 + np.clip(u, 0, s) % 2                # the cell inside its quad
    key = (tile * 512 + quad * 8 + sub).astype(np.float64)
    key[last] = np.inf
    return key


@pytest.mark.parametrize("d,size,P", [(2, (37, 50), 20011), (2, (256, 256), 200000), (3, (11, 20, 17), 30001)])
@pytest.mark.parametrize("mc,align", [(True, True), (False, True), (False, False)])
def test_sort_points_orders_by_cell(d, size, P, mc, align):
    pts = _points(P, d, seed=5 + d, spread=1.05)
    srt, perm = ops.sort_points(pts.to(DEV), size, 0, align, mc)
    torch.cuda.synchronize()
    srt, perm = srt.cpu(), perm.cpu()
    assert sorted(perm.tolist()) == list(range(P)), "perm is a permutation"
    assert torch.equal(srt, pts[perm]), "sorted_points = points[perm]"
    key = _key_reference(pts, size, 0, align, mc)
    ks = key[perm.numpy()]
    # ordered by key; the GPU's fused multiply-add may put a point that sits on a cell boundary to 1 ulp on the other side
    # of it than numpy does: allow a handful of such points, nothing else
    bad = int((np.diff(ks) < 0).sum())
    assert bad <= max(2, P // 20000), "%d inversions of the cell key" % bad
    same = np.diff(ks) == 0
    assert (np.diff(perm.numpy())[same] > 0).all(), "equal cells keep the caller's order (stable)"
    # the measure: an ordered set changes tile about once per occupied tile, an unordered one at almost every point
    ch_sorted = ops.points_tile_changes(srt.to(DEV), size, 0, align, mc)
    ch_random = ops.points_tile_changes(pts.to(DEV), size, 0, align, mc)
    ntiles = 1
    for s in size:
        ntiles *= s // 8 + 1
    assert ch_sorted <= ntiles + 2
    assert ch_random > 4 * ch_sorted or ch_random > P // 2


def _case(N, C, size, pts, seed, broadcast=False):
    g = torch.Generator().manual_seed(seed)
    P = pts.shape[0]
    inp = torch.rand((N, C) + tuple(size), generator=g)
    grid = pts.view(1, 1, P, 2)
    if not broadcast:
        grid = grid.repeat(N, 1, 1, 1).contiguous()
    oshape = (N, C, 1, P)
    gs = (grid.shape[0], 1, P, 2)
    return dict(inp=inp, grid=grid, gOut=torch.randn(oshape, generator=g), cI=torch.randn(inp.shape, generator=g),
                cG=torch.randn(gs, generator=g), hG=torch.randn(gs, generator=g), hO=torch.randn(oshape, generator=g))


def _stages(mod, t, off, pad, align, ke, mc, dev, **kw):
    x = {k: v.to(dev) for k, v in t.items()}
    off = off.to(dev)
    r = {}
    r["out"] = mod.forward(x["inp"], x["grid"], off, pad, align, ke, mc, **kw)
    r["gI"], r["gG"] = mod.backward(x["gOut"], x["inp"], x["grid"], off, pad, align, True, ke, mc, **kw)
    r["bbI"], r["bbG"], r["bbO"] = mod.backward_backward(x["cI"], x["cG"], x["inp"], x["grid"], x["gOut"], off, pad,
                                                         align, True, ke, mc, **kw)
    r["bbI0"], r["bbG0"], r["bbO0"] = mod.backward_backward(None, x["cG"], x["inp"], x["grid"], x["gOut"], off,
                                                            pad, align, False, ke, mc, **kw)
    r["k4I"], r["k4O"] = mod.backward_backward_backward(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], off, pad,
                                                        align, True, ke, mc, **kw)
    r["fI"], r["fO"] = mod.bbb_fused(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], x["hO"], off, pad, align, ke,
                                     mc, **kw)
    return r


COH_CASES = []
for _C in (4, 8, 16):
    for _ke, _pad, _align, _mc in ((0, 0, True, True), (2, 0, False, False), (1, 1, True, False), (0, 2, True, True),
                                   (2, 2, False, True), (0, 1, False, True)):
        COH_CASES.append((_C, _ke, _pad, _align, _mc))
COH_CASES += [(32, 0, 0, True, True), (32, 2, 1, False, False), (1, 0, 0, True, True), (3, 1, 2, True, True),
              (6, 2, 0, True, True), (12, 0, 0, True, True), (24, 0, 0, True, True)]


@pytest.mark.parametrize("C,ke,pad,align,mc", COH_CASES)
@pytest.mark.parametrize("how", ["sorted", "half", "random"])
def test_coherent_path_matches_cpu_oracle(C, ke, pad, align, mc, how):
    """Every backward stage that scatters, on the coherent kernels (forced: the hint is the caller's), against the oracle,
    for points in the helper's order, half-ordered and unordered -- the result must not depend on the order."""
    N, P, size = 5, 9001, (37, 50)
    pts = _order(_points(P, 2, seed=31 + C + ke), size, how, 0, True, mc, seed=77)
    t = _case(N, C, size, pts, seed=9000 + C + 10 * ke + pad)
    off = offsets(N, mc)

    class Oracle(object):       # the oracle knows nothing of contexts
        def __getattr__(self, name):
            fn = getattr(cs_oracle, name)
            return lambda *a, **k: fn(*a)
    want = _stages(Oracle(), t, off, pad, align, ke, mc, "cpu")
    ops.force_path(2)
    try:
        step = ops.StepContext(points_order="coherent")
        got = _stages(ops, t, off, pad, align, ke, mc, DEV, ctx=step)
        torch.cuda.synchronize()
        # the same stages when the input-shaped gradient is not wanted: nothing is scattered, the point results stay
        x = {k: v.to(DEV) for k, v in t.items()}
        nI, got["gG_only"] = ops.backward(x["gOut"], x["inp"], x["grid"], off.to(DEV), pad, align, False, ke, mc, ctx=step)
        nI2, got["bbG_only"], got["bbO_only"] = ops.backward_backward(None, x["cG"], x["inp"], x["grid"], x["gOut"], off.to(DEV),
                                                                      pad, align, False, ke, mc, ctx=step, want_grad_input=False)
        torch.cuda.synchronize()
        assert nI is None and nI2 is None
    finally:
        ops.force_path(0)
    want["gG_only"], want["bbG_only"], want["bbO_only"] = want["gG"], want["bbG0"], want["bbO0"]
    for k in want:
        assert_close(got[k], want[k], "coherent C=%d kernel=%d pad=%d align=%s mc=%s order=%s: %s"
                     % (C, ke, pad, align, mc, how, k))


@pytest.mark.parametrize("P", [1, 63, 64, 65, 255, 257, 4099])
def test_coherent_path_ragged_point_counts(P):
    """waves with a single live lane, exactly full waves, a last wave of one sample"""
    N, C, size = 3, 16, (24, 20)
    pts = _order(_points(max(P, 3), 2, seed=3)[:P].contiguous(), size, "sorted", 0, True, True, seed=1)
    t = _case(N, C, size, pts, seed=12)
    off = offsets(N, True)

    class Oracle(object):
        def __getattr__(self, name):
            fn = getattr(cs_oracle, name)
            return lambda *a, **k: fn(*a)
    want = _stages(Oracle(), t, off, 0, True, 0, True, "cpu")
    ops.force_path(2)
    try:
        got = _stages(ops, t, off, 0, True, 0, True, DEV, ctx=ops.StepContext(points_order="coherent"))
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for k in want:
        assert_close(got[k], want[k], "P=%d: %s" % (P, k))


def test_coherent_broadcast_grid_and_expanded_cotangents():
    """PIXEL's shapes: one (1,1,P,2) set of points for every table, cotangents expanded along n"""
    N, C, P, size = 6, 16, 8000, (48, 48)
    pts = _order(_points(P, 2, seed=9), size, "sorted", 0, True, True, seed=1)
    t = _case(N, C, size, pts, seed=4, broadcast=True)
    off = offsets(N, True)
    g1 = torch.randn(1, C, 1, P, generator=torch.Generator().manual_seed(2))
    t["gOut"] = g1.expand(N, C, 1, P)
    rep = dict(t)
    for k in ("grid", "cG", "hG"):
        rep[k] = t[k].repeat(N, 1, 1, 1).contiguous()
    rep["gOut"] = t["gOut"].contiguous()

    class Oracle(object):
        def __getattr__(self, name):
            fn = getattr(cs_oracle, name)
            return lambda *a, **k: fn(*a)
    want = _stages(Oracle(), rep, off, 0, True, 0, True, "cpu")
    ops.force_path(2)
    try:
        x = {k: v.to(DEV) for k, v in t.items()}
        x["gOut"] = g1.to(DEV).expand(N, C, 1, P)
        got = _stages(ops, x, off, 0, True, 0, True, DEV, ctx=ops.StepContext(points_order="coherent"))
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for k in want:
        w = want[k]
        if k in ("gG", "bbG", "bbG0"):       # gradient w.r.t. the shared points = sum over n
            w = w.sum(0, keepdim=True)
        assert_close(got[k], w, "coherent broadcast grid: %s" % k)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_coherent_half_streams_vs_oracle(dtype):
    """16-bit streams on the coherent kernels, against the ORACLE on the rounded inputs (fp32 arithmetic on both sides:
    the only difference is the rounding of the 16-bit outputs)"""
    N, C, P, size = 4, 16, 6000, (40, 40)
    pts = _order(_points(P, 2, seed=19), size, "sorted", 0, True, True, seed=1)
    t = _case(N, C, size, pts, seed=8)
    for k in ("gOut", "hO"):
        t[k] = t[k].to(dtype).float()           # the values a 16-bit caller holds
    off = offsets(N, True)

    class Oracle(object):
        def __getattr__(self, name):
            fn = getattr(cs_oracle, name)
            return lambda *a, **k: fn(*a)
    want = _stages(Oracle(), t, off, 0, True, 0, True, "cpu")
    x = dict(t)
    for k in ("gOut", "hO"):
        x[k] = t[k].to(dtype)
    ops.force_path(2)
    try:
        got = _stages(ops, x, off, 0, True, 0, True, DEV, ctx=ops.StepContext(points_order="coherent"))
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    tol16 = 2.0 ** (-10 if dtype == torch.float16 else -7)
    for k in want:
        out16 = got[k].dtype == dtype
        assert_close(got[k].float(), want[k], "coherent %s streams: %s" % (dtype, k), tol=tol16 if out16 else 1e-5)


def _coherent_calls():
    return sum(c[2] for c in ops.call_counts.values())


def test_auto_mode_decides_per_tensor_and_never_from_another_tensors_measurement():
    """'auto' runs the coherent kernels only on a grid tensor that was ITSELF measured as ordered (a stale 'coherent' on
    unordered points costs 20-40x): a new tensor takes the general path unless the last tensor of its signature was
    ordered, in which case the host waits for this tensor's own count; results agree on every path"""
    N, C, P, size = 4, 8, 40000, (64, 64)
    off = offsets(N, True).to(DEV)
    ops.points_order("auto")          # (also forgets every remembered tensor)
    ops.force_path(2)
    try:
        def tensors(how, seed):
            pts = _order(_points(P, 2, seed=seed), size, how, 0, True, True, seed=1)
            return {k: v.to(DEV) for k, v in _case(N, C, size, pts, seed=6).items()}

        def run(t, busy=False):
            before, waits = _coherent_calls(), ops.order_waits
            if busy:          # the GPU has ~10 ms of other work queued: a count asked for now cannot be back when the op looks
                torch.cuda._sleep(20_000_000)
            r = ops.backward(t["gOut"], t["inp"], t["grid"], off, 0, True, True, 0, True, ctx=ops.StepContext())
            torch.cuda.synchronize()
            return r, _coherent_calls() - before, ops.order_waits - waits

        srt, rnd, srt2 = tensors("sorted", 23), tensors("random", 23), tensors("sorted", 23)
        r1, coh, waits = run(srt, busy=True)
        assert (coh, waits) == (0, 0), "first sight of a tensor, nothing known about its signature: general path, no wait"
        r2, coh, waits = run(srt)
        assert (coh, waits) == (1, 0), "the same tensor again: its own measurement has arrived"
        r3, coh, waits = run(rnd, busy=True)
        assert (coh, waits) == (0, 1), "a NEW tensor after an ordered one: the host waits for ITS count -- unordered"
        r4, coh, waits = run(rnd)
        assert (coh, waits) == (0, 0)
        r5, coh, waits = run(srt2, busy=True)
        assert (coh, waits) == (0, 0), "the last tensor of the signature was unordered: nothing waits, general path"
        r6, coh, waits = run(srt2)
        assert (coh, waits) == (1, 0)
        # table 0 ordered, every other table not: the whole grid is measured, not table 0
        mixed = dict(srt)
        mixed["grid"] = torch.cat([srt["grid"][:1], rnd["grid"][1:]]).contiguous()
        for _ in range(3):
            _, coh, _w = run(mixed)
            assert coh == 0, "a grid whose table 0 alone is ordered must not run on the coherent kernels"
        for a, b in ((r1, r2), (r3, r4), (r5, r6), (r1, r5)):
            assert rel_err(a[0], b[0]) <= 1e-5 and rel_err(a[1], b[1]) <= 1e-5
    finally:
        ops.force_path(0)
        ops.points_order("auto")


def test_auto_mode_has_no_cliff():
    """no call in 'auto' mode costs more than 1.5x the general path: neither a grid whose first table alone is ordered, nor
    the first unordered tensor after a run of ordered ones (the reference has no order-dependent cliff, 2d.cu:464-505)"""
    N, C, P, size = 8, 16, 1 << 17, (128, 128)
    off = offsets(N, True).to(DEV)
    pts = _points(P, 2, seed=41)
    srt = ops.sort_points(pts.to(DEV), size)[0]
    g_rnd = pts.to(DEV).view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
    g_srt = srt.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
    g_mix = torch.cat([g_srt[:1], g_rnd[1:]]).contiguous()
    gen = torch.Generator().manual_seed(3)
    inp = torch.rand(N, C, *size, generator=gen).to(DEV)
    gOut = torch.randn(N, C, 1, P, generator=gen).to(DEV)

    def timed(grid, reps=1):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.backward(gOut, inp, grid, off, 0, True, True, 0, True, ctx=ops.StepContext())
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ops.points_order("random")
    timed(g_rnd)
    general = min(timed(g_rnd) for _ in range(3))
    ops.points_order("auto")
    try:
        worst = max(timed(g_mix) for _ in range(4))
        assert worst <= 1.5 * general, "table 0 ordered, the others not: %.3f ms vs %.3f ms general" % (worst, general)
        for _ in range(3):
            timed(g_srt)                                   # the signature's history now says 'ordered' ...
        assert any(e.decision for e in ops._order_known)
        fresh = g_rnd.clone()                              # ... and the next tensor is not
        t = timed(fresh)
        assert t <= 1.5 * general, "ordered -> unordered switch: %.3f ms vs %.3f ms general" % (t, general)
    finally:
        ops.points_order("auto")


def test_order_measurement_samples_the_whole_set():
    """a point set whose first half is in cell order and whose second half is not: a measurement on a prefix would call
    it coherent (and the coherent kernels would crawl through the second half); the sampled one must not"""
    size, P = (128, 128), 1 << 18
    pts = _points(P, 2, seed=31)
    srt, _ = ops.sort_points(pts.to(DEV), size)
    mixed = torch.cat([srt[: P // 2], pts[P // 2:].to(DEV)]).contiguous()
    lib = ops._lib.load()
    word = torch.zeros(1, dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream

    def sampled(t):
        ops._lib.check(lib.cs_points_tile_changes_sampled(2, t.data_ptr(), word.data_ptr(), P, 1, size[0], size[1], 0, 1, 1,
                                                          ops.ORDER_SAMPLE_SEGMENTS, st), "cs_points_tile_changes_sampled")
        return int(word.item())

    n = ops.ORDER_SAMPLE_SEGMENTS * 1024
    assert sampled(srt) * 256 <= n, "ordered set not recognised"
    assert sampled(mixed) * 256 > n, "half-ordered set taken for an ordered one"
    assert sampled(pts.to(DEV)) * 256 > n
    # all of a small set is looked at: the sample equals the full count
    small = srt[:5000].contiguous()
    ops._lib.check(lib.cs_points_tile_changes_sampled(2, small.data_ptr(), word.data_ptr(), 5000, 1, size[0], size[1], 0, 1, 1,
                                                      ops.ORDER_SAMPLE_SEGMENTS, st), "cs_points_tile_changes_sampled")
    assert int(word.item()) == ops.points_tile_changes(small, size)


def test_step_context_does_not_confuse_reallocated_cotangents():
    """VERDICT r2 / ADVICE r2 (high): the sorted copy of grad_output in the plan used to be remembered by address, version,
    shape and strides with no reference held; a freed cotangent's block is handed to the next tensor of the same size, and
    the later stage then streamed the FIRST tensor's rows.  Now the context holds what it remembers."""
    N, C, P, size = 3, 16, 30000, (40, 40)
    pts = _points(P, 2, seed=41)
    t = _case(N, C, size, pts, seed=3)
    off = offsets(N, True)
    x = {k: v.to(DEV) for k, v in t.items()}
    offd = off.to(DEV)
    ops.force_path(2)
    try:
        sc = ops.StepContext(points_order="random")
        g1 = torch.randn(N, C, 1, P, device=DEV)
        ops.backward(g1, x["inp"], x["grid"], offd, 0, True, True, 0, True, ctx=sc)       # leaves g1's sorted copy
        torch.cuda.synchronize()
        addr = g1.data_ptr()
        sc_drops = ops.StepContext(points_order="random")      # a context that does NOT hold g1, to provoke the re-use
        del g1
        g2 = torch.randn(N, C, 1, P, device=DEV)                # the allocator may hand out g1's block again
        if sc._sorted_go is not None:
            assert g2.data_ptr() != addr, "the context holds the tensor whose copy it remembers: its block cannot be re-used"
        want = cs_oracle.backward_backward(None, t["cG"], t["inp"], t["grid"], g2.cpu(), off, 0, True, False, 0, True)
        got = ops.backward_backward(None, x["cG"], x["inp"], x["grid"], g2, offd, 0, True, False, 0, True, ctx=sc)
        for a, b, nm in zip(got, want, ("gInput", "gGrid", "ggOut")):
            assert_close(a, b, "second backward with a new cotangent on a used context: %s" % nm)
        want = cs_oracle.bbb_fused(t["inp"], t["grid"], g2.cpu(), t["cG"], t["hG"], t["hO"], off, 0, True, 0, True)
        got = ops.bbb_fused(x["inp"], x["grid"], g2, x["cG"], x["hG"], x["hO"], offd, 0, True, 0, True, ctx=sc)
        for a, b, nm in zip(got, want, ("gInput", "ggOut")):
            assert_close(a, b, "third backward with a new cotangent on a used context: %s" % nm)
        # the same for the grid the plan belongs to: a new grid tensor is a new plan
        grid2 = (x["grid"] * 0.5).contiguous()
        want = cs_oracle.backward(g2.cpu(), t["inp"], grid2.cpu(), off, 0, True, True, 0, True)
        got = ops.backward(g2, x["inp"], grid2, offd, 0, True, True, 0, True, ctx=sc)
        for a, b, nm in zip(got, want, ("grad_input", "grad_grid")):
            assert_close(a, b, "backward with a new grid on a used context: %s" % nm)
        del sc_drops
    finally:
        ops.force_path(0)


def test_two_fp64_cotangents_on_one_sampler_call():
    """ADVICE r2 (high), the autograd form: fp64 cotangents are converted to fp32 temporaries on every call; two different
    ones of the same shape on one sampler call must not share a sorted copy"""
    from cosinesampler_amd import CosineSampler2d
    N, C, P, size = 2, 8, 40000, (32, 32)
    g = torch.Generator().manual_seed(5)
    cells = torch.rand((N, C) + size, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True)
    grid = (torch.rand(N, 1, P, 2, generator=g, dtype=torch.float64) * 2 - 1).to(DEV).requires_grad_(True)
    a = torch.randn(N, C, 1, P, generator=g, dtype=torch.float64).to(DEV)
    b = torch.randn(N, C, 1, P, generator=g, dtype=torch.float64).to(DEV)
    ops.force_path(2)
    try:
        res = {}
        for nm in ("shared", "separate"):
            out = CosineSampler2d.apply(cells, grid, "zeros", True, "cosine", True)
            ga = torch.autograd.grad(out, (cells, grid), a, create_graph=True)
            if nm == "separate":
                out = CosineSampler2d.apply(cells, grid, "zeros", True, "cosine", True)
            gb = torch.autograd.grad(out, (cells, grid), b, create_graph=True)
            la = (ga[1] ** 2).sum()
            lb = (gb[1] ** 2).sum()
            res[nm] = torch.autograd.grad(la, cells, retain_graph=True)[0], torch.autograd.grad(lb, cells)[0]
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for i in range(2):
        assert rel_err(res["shared"][i], res["separate"][i]) <= 1e-5


def test_exact_mixed_second_backward_then_third_on_a_shared_context():
    """ADVICE r2 (medium b): the '+mixed' second backward with grad_out_input runs on kernels that leave no sorted copy;
    a third backward on the same context must not be told there is one"""
    N, C, P, size = 3, 16, 20000, (36, 36)
    pts = _points(P, 2, seed=43)
    t = _case(N, C, size, pts, seed=13)
    off = offsets(N, True)
    x = {k: v.to(DEV) for k, v in t.items()}
    offd = off.to(DEV)
    ke = 0 | ops.EXACT_MIXED
    ops.force_path(2)
    try:
        sc = ops.StepContext(points_order="random")
        ops.backward(x["gOut"], x["inp"], x["grid"], offd, 0, True, False, ke, True, ctx=sc)
        ops.backward_backward(x["cI"], x["cG"], x["inp"], x["grid"], x["gOut"], offd, 0, True, True, ke, True, ctx=sc)
        got = ops.bbb_fused(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], x["hO"], offd, 0, True, ke, True, ctx=sc)
        ref = ops.bbb_fused(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], x["hO"], offd, 0, True, ke, True)
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for a, b, nm in zip(got, ref, ("gInput", "ggOut")):
        assert_close(a, b, "third backward after an exact second backward on one context: %s" % nm)


def test_misaligned_views_are_accepted():
    """The C ABI states an alignment contract (include/cosine_sampler.h) and refuses violations; the Python layer copies a
    contiguous view that starts at an odd place instead of handing it over"""
    N, C, P, size = 2, 3, 1001, (9, 7)
    g = torch.Generator().manual_seed(1)
    big = torch.rand(1 + N * C * 63, generator=g).to(DEV)
    inp = big[1:].view(N, C, *size)                      # 4-byte aligned only
    assert inp.data_ptr() % 16 != 0 and inp.is_contiguous()
    gbig = (torch.rand(1 + N * P * 2, generator=g) * 2 - 1).to(DEV)
    grid = gbig[1:].view(N, 1, P, 2)
    assert grid.data_ptr() % 8 != 0
    off = multicell_offset(N, True, DEV)
    out = ops.forward(inp, grid, off, 0, True, 0, True)
    want = cs_oracle.forward(inp.cpu().contiguous(), grid.cpu().contiguous(), off.cpu(), 0, True, 0, True)
    assert_close(out, want, "forward on misaligned views")
    lib = _lib.load()
    o = torch.empty(N, C, 1, P, device=DEV)
    rc = lib.cs2d_forward(inp.data_ptr(), grid.clone().data_ptr(), off.data_ptr(), o.data_ptr(), N, C, size[0], size[1], P,
                          0, 1, 0, 1, None, None, None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == -1, "the C ABI refuses a misaligned input (CS_ERR_INVALID), got %d" % rc


def test_full_size_ordered_points_at_the_headline_config():
    """BASELINE configs[1] sizes (N=16 C=16 256^2, P=2^20) on ordered points: the coherent kernels against the general
    path on the same inputs (two independent algorithms: run reduction on chip vs sort + records + walkers), and against
    size-independent properties: blending weights sum to one (sum over the nodes of grad_input = sum over the points of
    grad_output for points inside the table), and the order of the points cannot matter (the ordered run un-permuted
    equals the run on the points as drawn)."""
    N, C, H, P = 16, 16, 256, 1 << 20
    g = torch.Generator().manual_seed(21)
    cells = torch.rand(N, C, H, H, generator=g).to(DEV)
    xy = ((torch.rand(P, 2, generator=g) * 2 - 1) * 0.999).to(DEV)
    xy_s, perm = ops.sort_points(xy, (H, H))
    changes = ops.points_tile_changes(xy_s, (H, H))
    assert changes <= (H // 8 + 1) ** 2 + 2 and changes * 256 <= P
    off = multicell_offset(N, True, DEV)
    gOut = torch.randn(N, C, 1, P, generator=g).to(DEV)
    hO = torch.randn(N, C, 1, P, generator=g).to(DEV)
    cG = torch.randn(N, 1, P, 2, generator=g).to(DEV)
    hG = torch.randn(N, 1, P, 2, generator=g).to(DEV)

    def run(pts, go, ho, cg, hg, order):
        grid = pts.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
        sc = ops.StepContext(points_order=order)
        r = {}
        r["gI"], r["gG"] = ops.backward(go, cells, grid, off, 0, True, True, 0, True, ctx=sc)
        r["bbI"], r["bbG"], r["bbO"] = ops.backward_backward(None, cg, cells, grid, go, off, 0, True, False, 0, True, ctx=sc)
        r["fI"], r["fO"] = ops.bbb_fused(cells, grid, go, cg, hg, ho, off, 0, True, 0, True, ctx=sc)
        torch.cuda.synchronize()
        return r

    pm = perm.to(DEV)
    go_s, ho_s = gOut[..., pm].contiguous(), hO[..., pm].contiguous()
    cg_s, hg_s = cG[:, :, pm].contiguous(), hG[:, :, pm].contiguous()
    coh = run(xy_s, go_s, ho_s, cg_s, hg_s, "coherent")
    gen = run(xy_s, go_s, ho_s, cg_s, hg_s, "random")
    for k in coh:
        assert rel_err(coh[k], gen[k]) <= 1e-5, "coherent vs general path on ordered points: %s %.2e" % (k, rel_err(coh[k], gen[k]))
    del gen
    # weights sum to one: every point is inside the table here
    s_nodes = coh["gI"].double().sum((2, 3))
    s_pts = go_s.double().sum((2, 3))
    assert float((s_nodes - s_pts).abs().max()) <= 1e-4 * float(s_pts.abs().max())
    # the order cannot matter
    drawn = run(xy, gOut, hO, cG, hG, "random")
    for k in ("gI", "bbI", "fI"):
        assert rel_err(coh[k], drawn[k]) <= 1e-5, "ordered vs as drawn: %s" % k
    inv = torch.empty_like(pm)
    inv[pm] = torch.arange(P, device=DEV)
    for k in ("gG", "bbG"):
        assert rel_err(coh[k][:, :, inv], drawn[k]) <= 1e-5, "ordered vs as drawn: %s" % k
    for k in ("bbO", "fO"):
        assert rel_err(coh[k][..., inv], drawn[k]) <= 1e-5, "ordered vs as drawn: %s" % k


def test_ordered_points_slice_vs_oracle_at_full_table_size():
    """... and against the ORACLE on what it can do in seconds: one table of the headline size (256^2, C=16) with 2^16
    ordered points that all fall into a 24 x 24-cell corner of it (16 per cell, the headline's density)"""
    N, C, H, P = 2, 16, 256, 1 << 16
    g = torch.Generator().manual_seed(22)
    cells = torch.rand(N, C, H, H, generator=g)
    xy = (torch.rand(P, 2, generator=g) * 2 - 1) * (48.0 / 254.0) - 0.8
    xy_s, _ = ops.sort_points(xy.to(DEV), (H, H))
    t = _case(N, C, (H, H), xy_s.cpu(), seed=5)
    t["inp"] = cells
    off = offsets(N, True)

    class Oracle(object):
        def __getattr__(self, name):
            fn = getattr(cs_oracle, name)
            return lambda *a, **k: fn(*a)
    want = _stages(Oracle(), t, off, 0, True, 0, True, "cpu")
    got = _stages(ops, t, off, 0, True, 0, True, DEV, ctx=ops.StepContext(points_order="coherent"))
    torch.cuda.synchronize()
    for k in want:
        assert_close(got[k], want[k], "ordered points on a 256^2 table vs oracle: %s" % k)


@pytest.mark.parametrize("order", ["drawn", "sorted"])
def test_full_size_helmholtz_autograd(order):
    """BASELINE configs[2] at its full size through torch.autograd (N=16 C=16 256^2, P=2^20): the PIXEL-style Helmholtz
    step (reference test/test_2d.py:36-52 pattern) with the points as drawn and ordered by cell (the coherent kernels,
    chosen by the op's own measurement).  Checked: (a) a (1,1,P,2) broadcast grid gives what the repeated grid gives;
    (b) on a 4096-point slice, u, u_x, u_xx and d(loss restricted to the slice)/d cells against the exact-derivative
    composite (oracle/composite.py, float64 on the GPU: the checker, not the product) evaluated on the slice alone."""
    from cosinesampler_amd import CosineSampler2d, CosineSampler2dSum
    from oracle import composite
    N, C, H, P, K = 16, 16, 256, 1 << 20, 4096
    g = torch.Generator().manual_seed(31)
    cells0 = torch.rand(N, C, H, H, generator=g).to(DEV)
    W1 = (torch.randn(16, C, generator=g) * 0.5).to(DEV)
    W2 = (torch.randn(1, 16, generator=g) * 0.5).to(DEV)
    # half a lattice step away from every cell boundary in every table (k'' jumps there: DESIGN.md section 2)
    lat = (torch.randint(0, 254 * 16, (P, 2), generator=g).float() + 0.5) / (254 * 16)
    xy = (lat * 2 - 1).to(DEV)
    if order == "sorted":
        xy, _ = ops.sort_points(xy, (H, H))
    sel = torch.randperm(P, generator=g)[:K].to(DEV)
    mask = torch.zeros(P, 1, device=DEV)
    mask[sel] = 1.0

    def step(bc, summed=False):
        cells = cells0.clone().requires_grad_(True)
        x = xy[:, :1].clone().requires_grad_(True)
        y = xy[:, 1:].clone().requires_grad_(True)
        ones = torch.ones(P, 1, device=DEV)
        grid = torch.cat([x, y], -1).view(1, 1, P, 2)
        if summed:          # (c) sampler(...).sum(0) as one op: on ordered points the summing kernels, else plain op + sums
            feat = CosineSampler2dSum.apply(cells, grid, "zeros", True, "cosine", True)[0]
        else:
            if not bc:
                grid = grid.repeat(N, 1, 1, 1)
            feat = CosineSampler2d.apply(cells, grid, "zeros", True, "cosine", True).sum(0)
        u = torch.tanh(feat.view(C, -1).t() @ W1.t()) @ W2.t()
        u_x, u_y = torch.autograd.grad(u, (x, y), ones, create_graph=True)
        (u_xx,) = torch.autograd.grad(u_x, x, ones, create_graph=True)
        (u_yy,) = torch.autograd.grad(u_y, y, ones, create_graph=True)
        loss = torch.sum(mask * (u_xx + u_yy + 4.0 * u) ** 2) / K
        (gc,) = torch.autograd.grad(loss, cells)
        return dict(u=u.detach(), u_x=u_x.detach(), u_xx=u_xx.detach(), u_yy=u_yy.detach(), gc=gc)

    ops.points_order("auto")    # (forgets every remembered tensor and history)
    for _ in range(2):          # the order measurement arrives a call late
        step(False)
        torch.cuda.synchronize()
    rep = step(False)
    bcr = step(True)
    smd = step(True, summed=True)
    torch.cuda.synchronize()
    if order == "sorted":
        assert any(e.decision for e in ops._order_known), "ordered points were not recognised"
        assert ops.sum_over_n_fused(cells0, xy.view(1, 1, P, 2), 0, True, True), "the summing kernels were not used"
    for k in rep:
        assert rel_err(bcr[k], rep[k]) <= 2e-5, "broadcast vs repeated grid: %s %.2e" % (k, rel_err(bcr[k], rep[k]))
        assert rel_err(smd[k], rep[k]) <= 2e-5, "summed op vs repeat + sum: %s %.2e" % (k, rel_err(smd[k], rep[k]))
    # the slice against the composite in float64
    cells = cells0.double().clone().requires_grad_(True)
    x = xy[sel, :1].double().clone().requires_grad_(True)
    y = xy[sel, 1:].double().clone().requires_grad_(True)
    ones = torch.ones(K, 1, device=DEV, dtype=torch.float64)
    grid = torch.cat([x, y], -1).view(1, 1, K, 2).repeat(N, 1, 1, 1)
    val = composite.grid_sample_nd(cells, grid, "cosine", True, True)
    u = torch.tanh(val.sum(0).view(C, -1).t() @ W1.double().t()) @ W2.double().t()
    u_x, u_y = torch.autograd.grad(u, (x, y), ones, create_graph=True)
    (u_xx,) = torch.autograd.grad(u_x, x, ones, create_graph=True)
    (u_yy,) = torch.autograd.grad(u_y, y, ones, create_graph=True)
    loss = torch.sum((u_xx + u_yy + 4.0 * u) ** 2) / K
    (gc,) = torch.autograd.grad(loss, cells)
    for nm, got, want in (("u", rep["u"][sel], u), ("u_x", rep["u_x"][sel], u_x), ("u_xx", rep["u_xx"][sel], u_xx),
                          ("u_yy", rep["u_yy"][sel], u_yy), ("d loss / d cells", rep["gc"], gc)):
        # Bounds of THIS pipeline at THIS table size, not of the op (which is held to 1e-5 against the oracle, bit-identical
        # source index, everywhere else): the checker works in float64 from the same fp32 coordinates, the op in fp32, and
        # the source index i ~ 256 carries one fp32 ulp = 1.5e-5 cells, which k'' = (pi^2/2) cos(pi t) turns into ~2e-4
        # relative on second derivatives.  u and u_x: the reference's own rtol
        # (test/test_2d.py:244); second derivatives and what is built from them: 5e-4.
        tol = 1e-4 if nm in ("u", "u_x") else 5e-4
        assert rel_err(got, want.detach()) <= tol, "full-size Helmholtz (%s points) vs composite: %s %.2e" % (
            order, nm, rel_err(got, want.detach()))


def test_plan_cache_reuses_the_plan_of_an_unchanged_grid():
    """ops.plan_cache: a second step with the SAME grid tensor finds the plan; an in-place change of the grid, another
    tensor at the same address or a switched-off cache do not.  Results equal the uncached ones bit for bit."""
    N, C, P, size = 4, 16, 70000, (64, 64)
    t = _case(N, C, size, _points(P, 2, seed=21), seed=8)
    off = offsets(N, True).to(DEV)
    x = {k: v.to(DEV) for k, v in t.items()}

    def step():
        sc = ops.StepContext(points_order="random")
        gI, gG = ops.backward(x["gOut"], x["inp"], x["grid"], off, 0, True, True, 0, True, ctx=sc)
        return sc, gI, gG
    ops.force_path(2)
    try:
        _, gI0, gG0 = step()
        assert ops.plan_cache(1) == 1
        sc1, gI1, gG1 = step()
        sc2, gI2, gG2 = step()
        assert sc1._pe is not None and sc2._pe is sc1._pe, "the second step re-uses the first one's plan"
        assert torch.equal(gG1, gG0) and torch.equal(gG2, gG0)
        assert_close(gI2, gI0, "grad_input from a cached plan")
        x["grid"].mul_(0.5)                                   # in-place update: another version of the same tensor
        sc3, gI3, _ = step()
        assert sc3._pe is not sc1._pe, "an in-place change of the grid invalidates its plan"
        want = cs_oracle.backward(t["gOut"], t["inp"], t["grid"] * 0.5, offsets(N, True), 0, True, True, 0, True)
        assert_close(gI3, want[0], "grad_input after the in-place change")
        ops.plan_cache(0)
        sc4, _, _ = step()
        sc5, _, _ = step()
        assert sc5._pe is not sc4._pe
    finally:
        ops.plan_cache(0)
        ops.force_path(0)


# ---- CS_SUM_OVER_N: the PIXEL pattern (one set of points and cotangents, per-point results summed over the tables) -------
def _sum_n_case(N, C, size, P, seed, order="sorted"):
    pts = _order(_points(P, 2, seed=seed), size, order, 0, True, True, seed=seed + 1)
    g = torch.Generator().manual_seed(seed + 7)
    return dict(inp=torch.rand((N, C) + tuple(size), generator=g), grid=pts.view(1, 1, P, 2).contiguous(),
                gOut=torch.randn(1, C, 1, P, generator=g), cG=torch.randn(1, 1, P, 2, generator=g),
                hG=torch.randn(1, 1, P, 2, generator=g), hO=torch.randn(1, C, 1, P, generator=g))


def _sum_n_oracle(t, off, ke, mc):
    """the reference's way: repeated grid, expanded cotangents, sums over n afterwards (test/test_2d.py:38, :51)"""
    N = t["inp"].shape[0]
    rep = lambda x: x.repeat((N,) + (1,) * (x.dim() - 1)).contiguous()
    grid, gOut, cG, hG, hO = (rep(t[k]) for k in ("grid", "gOut", "cG", "hG", "hO"))
    s0 = lambda x: x.sum(0, keepdim=True)
    r = {}
    r["out"] = s0(cs_oracle.forward(t["inp"], grid, off, 0, True, ke, mc))
    gI, gG = cs_oracle.backward(gOut, t["inp"], grid, off, 0, True, True, ke, mc)
    r["gI"], r["gG"] = gI, s0(gG)
    bI, bG, bO = cs_oracle.backward_backward(None, cG, t["inp"], grid, gOut, off, 0, True, False, ke, mc)
    r["bbI"], r["bbG"], r["bbO"] = bI, s0(bG), s0(bO)
    fI, fO = cs_oracle.bbb_fused(t["inp"], grid, gOut, cG, hG, hO, off, 0, True, ke, mc)
    r["fI"], r["fO"] = fI, s0(fO)
    kI, kO = cs_oracle.backward_backward_backward(t["inp"], grid, gOut, cG, hG, off, 0, True, True, ke, mc)   # no gOutggOut
    r["kI"], r["kO"] = kI, s0(kO)
    return r


def _sum_n_gpu(t, off, ke, mc, ctx):
    x = {k: v.to(DEV) for k, v in t.items()}
    o = off.to(DEV)
    r = {}
    r["out"] = ops.forward_sum_n(x["inp"], x["grid"], o, 0, True, ke, mc, ctx=ctx)
    r["gI"], r["gG"] = ops.backward_sum_n(x["gOut"], x["inp"], x["grid"], o, 0, True, True, ke, mc, ctx=ctx)
    none_gi, gG0 = ops.backward_sum_n(x["gOut"], x["inp"], x["grid"], o, 0, True, False, ke, mc, ctx=ctx)
    assert none_gi is None and rel_err(gG0, r["gG"]) <= 1e-6
    r["bbI"], r["bbG"], r["bbO"] = ops.backward_backward_sum_n(x["cG"], x["inp"], x["grid"], x["gOut"], o, 0, True, ke, mc, ctx=ctx)
    n2, bG0, bO0 = ops.backward_backward_sum_n(x["cG"], x["inp"], x["grid"], x["gOut"], o, 0, True, ke, mc, ctx=ctx,
                                               want_grad_input=False)
    assert n2 is None and rel_err(bG0, r["bbG"]) <= 1e-6 and rel_err(bO0, r["bbO"]) <= 1e-6
    r["fI"], r["fO"] = ops.bbb_fused_sum_n(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], x["hO"], o, 0, True, ke, mc, ctx=ctx)
    r["kI"], r["kO"] = ops.bbb_fused_sum_n(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], None, o, 0, True, ke, mc, ctx=ctx)
    torch.cuda.synchronize()
    return r


_SUM_N_SHAPES = [(4, 16, (64, 64), 40000), (3, 8, (37, 50), 20011), (16, 4, (32, 32), 30000), (2, 32, (40, 40), 9000),
                 (5, 3, (20, 33), 5000), (4, 12, (48, 48), 12345), (3, 16, (16, 16), 3), (2, 8, (16, 16), 63),
                 (2, 8, (16, 16), 129), (6, 16, (24, 24), 257)]
# (unordered points on the coherent kernels are correct but slow -- a window reload per sample: the smaller shapes cover them)
_SUM_N_CASES = [(sh, "sorted") for sh in _SUM_N_SHAPES] + [(sh, "random") for sh in _SUM_N_SHAPES if sh[3] <= 5000]


@pytest.mark.parametrize("shape,order", _SUM_N_CASES)
@pytest.mark.parametrize("ke,mc", [(0, True), (2, True), (1, False)])
def test_sum_over_n_kernels_match_the_cpu_oracle(shape, order, ke, mc):
    """the summing kernels (forced on: ops.points_order('coherent'); they are correct for any order) against the oracle run
    the reference's way -- repeated grid, expanded cotangents, sums over n afterwards"""
    N, C, size, P = shape
    t = _sum_n_case(N, C, size, P, seed=100 + C + P % 97, order=order)
    off = offsets(N, mc)
    want = _sum_n_oracle(t, off, ke, mc)
    ops.points_order("coherent")
    ops.force_path(2)
    try:
        assert ops.sum_over_n_fused(t["inp"].to(DEV), t["grid"].to(DEV), 0, True, mc), "the summing kernels do not apply"
        got = _sum_n_gpu(t, off, ke, mc, ops.StepContext())
        got_noctx = _sum_n_gpu(t, off, ke, mc, None)
    finally:
        ops.force_path(0)
        ops.points_order("auto")
    for k in want:
        assert_close(got[k], want[k], "sum over n (%s points) %s" % (order, k))
        assert_close(got_noctx[k], want[k], "sum over n, no context (%s points) %s" % (order, k))


@pytest.mark.parametrize("why", ["unordered", "border", "align_false", "3d", "one_table", "bf16"])
def test_sum_over_n_falls_back_to_the_plain_op_and_sums(why):
    """where the summing kernels do not apply the *_sum_n functions give the same values from the plain op + torch sums"""
    N, C, size, P, pad, align, ke, mc = 4, 8, (32, 32), 20000, 0, True, 0, True
    t = _sum_n_case(N, C, size, P, seed=9)
    if why == "one_table":
        t["inp"] = t["inp"][:1].contiguous()
        N = 1
    off = offsets(N, mc)
    x = {k: v.to(DEV) for k, v in t.items()}
    o = off.to(DEV)
    ops.points_order("random" if why == "unordered" else "coherent")
    ops.force_path(2)
    try:
        if why == "border":
            pad = 1
        if why == "align_false":
            align = False
        if why == "3d":
            g = torch.Generator().manual_seed(3)
            # (3D tables of up to 16 channels have their own summing mode since round 4 -- tests/test_sum_op_gpu.py; wider ones
            # run in channel ranges through the plain op)
            x = dict(inp=torch.rand(3, 20, 8, 9, 10, generator=g).to(DEV), grid=(torch.rand(1, 1, 1, 3000, 3, generator=g) * 2 - 1).to(DEV),
                     gOut=torch.randn(1, 20, 1, 1, 3000, generator=g).to(DEV))
            o = offsets(3, mc).to(DEV)
            N = 3
        if why == "bf16":
            x["gOut"] = x["gOut"].bfloat16()
        # (unordered points are put into cell order inside the op since round 4: the summing kernels do run for them)
        assert not ops.sum_over_n_fused(x["inp"], x["grid"], pad, align, mc) or why in ("bf16", "unordered")
        if why == "unordered":
            assert ops.sum_over_n_mode(x["inp"], x["grid"], pad, align, mc) == "sorted"
        out = ops.forward_sum_n(x["inp"], x["grid"], o, pad, align, ke, mc)
        gI, gG = ops.backward_sum_n(x["gOut"], x["inp"], x["grid"], o, pad, align, True, ke, mc)
        rep = x["grid"].repeat((N,) + (1,) * (x["grid"].dim() - 1)).contiguous() if N > 1 else x["grid"]
        ops.points_order("random")
        out_p = ops.forward(x["inp"], rep, o, pad, align, ke, mc).sum(0, keepdim=True)
        gI_p, gG_p = ops.backward(x["gOut"].expand((N,) + tuple(x["gOut"].shape[1:])).contiguous(), x["inp"], rep, o, pad, align,
                                  True, ke, mc)
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
        ops.points_order("auto")
    tol = 1e-2 if why == "bf16" else 1e-5
    assert out.shape[0] == 1 and gG.shape[0] == 1 and gI.shape == x["inp"].shape
    assert_close(out, out_p, "fallback forward (%s)" % why, tol=1e-5)
    assert_close(gI.float(), gI_p.float(), "fallback grad_input (%s)" % why, tol=tol)
    assert_close(gG.float(), gG_p.sum(0, keepdim=True).float(), "fallback grad_grid (%s)" % why, tol=tol)

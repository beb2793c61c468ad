"""GPU (MI355X): CosineSampler{2,3}dSum (the PIXEL pattern sampler(cells, grid.repeat(N,..)).sum(0) as one op, reference
test/test_2d.py:38, :51) against the plain op followed by the sum, through torch.autograd."""
import pytest
import torch

from cosinesampler_amd import CosineSampler2d, CosineSampler2dSum, CosineSampler3d, CosineSampler3dSum, ops
from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("d,kernel", [(2, "cosine+mixed"), (2, "smooth-step+mixed"), (3, "cosine+mixed")])
def test_summed_op_returns_the_third_order_grid_gradient_of_mixed_kernels(d, kernel):
    """'+mixed' kernels return d/dgrid at third order (u_xxx, u_xxy; not in the reference, modules_2d.py:111): the summed op
    must return what the plain op + sum returns"""
    N, C, S, P = 3, 4, 12, 5000
    g = torch.Generator().manual_seed(17 + d)
    cells0 = torch.rand((N, C) + (S,) * d, generator=g).to(DEV)
    pts0 = (torch.rand((1,) * d + (P, d), generator=g) * 1.9 - 0.95).to(DEV)
    w = torch.randn((1, C) + (1,) * (d - 1) + (P,), generator=g).to(DEV)
    v1 = torch.randn(pts0.shape, generator=g).to(DEV)
    v2 = torch.randn(pts0.shape, generator=g).to(DEV)
    Sum, Plain = (CosineSampler2dSum, CosineSampler2d) if d == 2 else (CosineSampler3dSum, CosineSampler3d)

    def third(summed):
        cells = cells0.clone().requires_grad_(True)
        pts = pts0.clone().requires_grad_(True)
        if summed:
            feat = Sum.apply(cells, pts, "zeros", True, kernel, True)
        else:
            feat = Plain.apply(cells, pts.repeat((N,) + (1,) * (d + 1)), "zeros", True, kernel, True).sum(0, keepdim=True)
        (g1,) = torch.autograd.grad((feat * w).sum(), pts, create_graph=True)
        (g2,) = torch.autograd.grad((g1 * v1).sum(), pts, create_graph=True)
        g3, gc = torch.autograd.grad((g2 * v2).sum(), (pts, cells))
        return g1.detach(), g2.detach(), g3, gc

    ops.force_path(2)
    try:
        a, b = third(True), third(False)
    finally:
        ops.force_path(0)
    torch.cuda.synchronize()
    for x, y, nm in zip(a, b, ("first", "second", "third-order d/dgrid", "d/dcells")):
        assert x is not None, nm
        assert_close(x, y, "summed op vs plain op + sum: %s" % nm, 2e-5)


def _oracle_sum_n(t, off, ke, N):
    """the reference's way of writing the PIXEL pattern: repeat the points, expand the cotangents, sum afterwards"""
    from oracle import cs_oracle
    rep = lambda g: g.repeat((N,) + (1,) * (g.dim() - 1)).contiguous()
    exp = lambda g: g.expand((N,) + tuple(g.shape[1:])).contiguous()
    grid = rep(t["grid"])
    r = {}
    r["out"] = cs_oracle.forward(t["inp"], grid, off, 0, True, ke, True).sum(0, keepdim=True)
    gI, gG = cs_oracle.backward(exp(t["gOut"]), t["inp"], grid, off, 0, True, True, ke, True)
    r["gI"], r["gG"] = gI, gG.sum(0, keepdim=True)
    bI, bG, bO = cs_oracle.backward_backward(None, rep(t["cG"]), t["inp"], grid, exp(t["gOut"]), off, 0, True, False, ke, True)
    r["bbI"], r["bbG"], r["bbO"] = bI, bG.sum(0, keepdim=True), bO.sum(0, keepdim=True)
    tI, tO = cs_oracle.bbb_fused(t["inp"], grid, exp(t["gOut"]), rep(t["cG"]), rep(t["hG"]), exp(t["hO"]), off, 0, True, ke, True)
    r["tI"], r["tO"] = tI, tO.sum(0, keepdim=True)
    return r


@pytest.mark.parametrize("N,C,size,P,ke", [(4, 16, (48, 40), 30011, 0), (3, 5, (64, 64), 70001, 2), (16, 32, (16, 16), 20000, 0)])
def test_summed_op_orders_drawn_points_inside_the_op(N, C, size, P, ke):
    """points in the order they were drawn: the summed op sorts them once per step (ops.sum_n_sorts), runs the summing
    kernels, and hands every per-point result back in the caller's order -- against the oracle run the reference's way"""
    from helpers import offsets
    g = torch.Generator().manual_seed(41 + C)
    t = dict(inp=torch.rand((N, C) + size, generator=g), grid=torch.rand(1, 1, P, 2, generator=g) * 2.1 - 1.05,
             gOut=torch.randn(1, C, 1, P, generator=g), hO=torch.randn(1, C, 1, P, generator=g),
             cG=torch.randn(1, 1, P, 2, generator=g), hG=torch.randn(1, 1, P, 2, generator=g))
    off = offsets(N, True)
    want = _oracle_sum_n(t, off, ke, N)
    x = {k: v.to(DEV) for k, v in t.items()}
    o = off.to(DEV)
    ops.points_order("random")
    ops.force_path(2)
    try:
        sorts = ops.sum_n_sorts
        sc = ops.StepContext()
        assert ops.sum_over_n_mode(x["inp"], x["grid"], 0, True, True, sc) == "sorted"
        got = dict(out=ops.forward_sum_n(x["inp"], x["grid"], o, 0, True, ke, True, ctx=sc))
        got["gI"], got["gG"] = ops.backward_sum_n(x["gOut"], x["inp"], x["grid"], o, 0, True, True, ke, True, ctx=sc)
        got["bbI"], got["bbG"], got["bbO"] = ops.backward_backward_sum_n(x["cG"], x["inp"], x["grid"], x["gOut"], o, 0, True, ke,
                                                                           True, ctx=sc)
        got["tI"], got["tO"] = ops.bbb_fused_sum_n(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], x["hO"], o, 0, True, ke,
                                                    True, ctx=sc)
        torch.cuda.synchronize()
        assert ops.sum_n_sorts - sorts == 1, "one ordering of the points per step"
    finally:
        ops.force_path(0)
        ops.points_order("auto")
    for k in want:
        assert got[k].shape == want[k].shape, k
        assert_close(got[k], want[k], "summed op on drawn points: %s" % k)


SUM3_CASES = [   # N, C, (D, H, W), P, kernel, pad, align, multicell, force (2: plan + tile / cell scatter, 3: row atomics)
    (3, 8, (20, 9, 37), 30011, 2, 0, True, True, 2),     # the config-3 kind: sparse table, tile scatter
    (4, 8, (6, 9, 7), 5000, 0, 0, True, True, 2),        # crowded: wave-per-cell scatter
    (3, 5, (33, 5, 17), 20000, 1, 1, False, False, 2),   # channels padded to 8, border padding, align_corners = False
    (2, 16, (12, 6, 40), 25001, 2, 2, True, False, 2),   # 4 quads, reflection
    (5, 3, (7, 21, 18), 301, 0, 0, True, True, 3),       # row atomics, a last wave that is mostly idle
    (16, 4, (8, 8, 8), 63, 0, 0, True, True, 3),         # less than one wave of points, many tables
    (3, 8, (20, 9, 37), 30011, 2, 0, True, True, 3),     # the first case by row atomics
]


@pytest.mark.parametrize("N,C,size,P,ke,pad,align,mc,force", SUM3_CASES)
@pytest.mark.parametrize("shared", [False, True])
def test_summed_op_3d_walks_the_tables_per_point(N, C, size, P, ke, pad, align, mc, force, shared):
    """3D (round 4): CS_SUM_OVER_N on the channels-last point kernels -- one thread per point walks the N tables, per-point
    results summed in registers, the input-shaped gradients through the plain op's scatter -- against the oracle run the
    reference's way (repeat, expand, sum), on every scatter way, with and without the step's shared plan / table copy"""
    from helpers import offsets
    from oracle import cs_oracle
    g = torch.Generator().manual_seed(4100 + C + P)
    t = dict(inp=torch.rand((N, C) + size, generator=g), grid=torch.rand(1, 1, 1, P, 3, generator=g) * 2.2 - 1.1,
             gOut=torch.randn(1, C, 1, 1, P, generator=g), hO=torch.randn(1, C, 1, 1, P, generator=g),
             cG=torch.randn(1, 1, 1, P, 3, generator=g), hG=torch.randn(1, 1, 1, P, 3, generator=g))
    off = offsets(N, mc)
    rep = lambda x: x.repeat((N,) + (1,) * (x.dim() - 1)).contiguous()
    exp = lambda x: x.expand((N,) + tuple(x.shape[1:])).contiguous()
    grid = rep(t["grid"])
    want = {}
    want["out"] = cs_oracle.forward(t["inp"], grid, off, pad, align, ke, mc).sum(0, keepdim=True)
    want["gI"], gG = cs_oracle.backward(exp(t["gOut"]), t["inp"], grid, off, pad, align, True, ke, mc)
    want["bbI"], bG, bO = cs_oracle.backward_backward(None, rep(t["cG"]), t["inp"], grid, exp(t["gOut"]), off, pad, align,
                                                      False, ke, mc)
    want["tI"], tO = cs_oracle.bbb_fused(t["inp"], grid, exp(t["gOut"]), rep(t["cG"]), rep(t["hG"]), exp(t["hO"]), off, pad,
                                         align, ke, mc)
    want["gG"], want["bbG"], want["bbO"], want["tO"] = (v.sum(0, keepdim=True) for v in (gG, bG, bO, tO))
    x = {k: v.to(DEV) for k, v in t.items()}
    o = off.to(DEV)
    ops.force_path(force)
    try:
        sc = ops.StepContext() if shared else None
        assert ops.sum_over_n_mode(x["inp"], x["grid"], pad, align, mc, sc) == "kernels"
        got = dict(out=ops.forward_sum_n(x["inp"], x["grid"], o, pad, align, ke, mc, ctx=sc))
        got["gI"], got["gG"] = ops.backward_sum_n(x["gOut"], x["inp"], x["grid"], o, pad, align, True, ke, mc, ctx=sc)
        got["bbI"], got["bbG"], got["bbO"] = ops.backward_backward_sum_n(x["cG"], x["inp"], x["grid"], x["gOut"], o, pad, align,
                                                                           ke, mc, ctx=sc)
        got["tI"], got["tO"] = ops.bbb_fused_sum_n(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], x["hO"], o, pad, align, ke,
                                                    mc, ctx=sc)
        nI, g2 = ops.backward_sum_n(x["gOut"], x["inp"], x["grid"], o, pad, align, False, ke, mc, ctx=sc)
        torch.cuda.synchronize()
        assert nI is None and torch.equal(g2, got["gG"])
    finally:
        ops.force_path(0)
    for k in want:
        assert got[k].shape == want[k].shape, k
        assert_close(got[k], want[k], "3D summed op N=%d C=%d %s P=%d force=%d shared=%s: %s" % (N, C, size, P, force, shared, k))


def test_summed_op_3d_through_autograd_matches_the_plain_op():
    """CosineSampler3dSum through torch.autograd (first, second, third order) against CosineSampler3d on repeated points + sum"""
    N, C, S, P = 4, 8, 24, 40000
    g = torch.Generator().manual_seed(99)
    cells0 = torch.rand((N, C, S, S, S), generator=g).to(DEV)
    pts0 = (torch.rand((1, 1, 1, P, 3), generator=g) * 2 - 1).to(DEV)
    w = torch.randn((1, C, 1, 1, P), generator=g).to(DEV)
    v1, v2 = torch.randn(pts0.shape, generator=g).to(DEV), torch.randn(pts0.shape, generator=g).to(DEV)

    def run(summed):
        cells = cells0.clone().requires_grad_(True)
        pts = pts0.clone().requires_grad_(True)
        if summed:
            feat = CosineSampler3dSum.apply(cells, pts, "zeros", True, "smooth-step", True)
        else:
            feat = CosineSampler3d.apply(cells, pts.repeat(N, 1, 1, 1, 1), "zeros", True, "smooth-step", True).sum(0, keepdim=True)
        (g1,) = torch.autograd.grad((feat * w).sum(), pts, create_graph=True)
        (g2,) = torch.autograd.grad((g1 * v1).sum(), pts, create_graph=True)
        loss = (feat * w).sum() + (g1 * v1).sum() + (g2 * v2).sum()
        (gc,) = torch.autograd.grad(loss, cells)
        return feat.detach(), g1.detach(), g2.detach(), gc

    calls = dict((k, list(v)) for k, v in ops.call_counts.items())
    a = run(True)
    b = run(False)
    torch.cuda.synchronize()
    for x, y, nm in zip(a, b, ("features", "first", "second", "d/dcells")):
        assert_close(x, y, "3D summed op vs plain op + sum through autograd: %s" % nm, 2e-5)

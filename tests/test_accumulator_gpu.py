"""GPU (MI355X): the step accumulator (cs_cotangent_layout.accumulate_grad_input, ops.StepContext(accumulate=True)).

A training step wants the SUM of the input-shaped gradients of its backward stages (what autograd accumulates into
cells.grad, reference modules_2d.py:44, :74, :111 return them one by one): with an accumulating context the stages ADD into
one buffer -- NCHW for the walker / wave-per-cell paths, channels-last for the coherent-points kernels -- and
grad_input_sum() hands over the total.  Checked against the sum of the CPU oracle's three gradients, and against the same
stages run one by one.  Tolerance: helpers.REL_TOL (1e-5 relative)."""
import pytest
import torch

from cosinesampler_amd import _lib, ops
from helpers import assert_close, offsets
from oracle import cs_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(N, C, size, P, seed, d=2):
    g = torch.Generator().manual_seed(seed)
    inp = torch.rand((N, C) + tuple(size), generator=g)
    grid = torch.rand((N,) + (1,) * (d - 1) + (P, d), generator=g) * 2.2 - 1.1
    oshape = (N, C) + (1,) * (d - 1) + (P,)
    return dict(inp=inp, grid=grid, gOut=torch.randn(oshape, generator=g), cG=torch.randn(grid.shape, generator=g),
                hG=torch.randn(grid.shape, generator=g), hO=torch.randn(oshape, generator=g),
                cI=torch.randn(inp.shape, generator=g))


def _oracle_sum(t, off, pad, align, ke, mc, with_cI=False):
    gI = cs_oracle.backward(t["gOut"], t["inp"], t["grid"], off, pad, align, True, ke, mc)[0]
    bbI = cs_oracle.backward_backward(t["cI"] if with_cI else None, t["cG"], t["inp"], t["grid"], t["gOut"], off, pad, align,
                                      with_cI, ke, mc)[0]
    tI = cs_oracle.bbb_fused(t["inp"], t["grid"], t["gOut"], t["cG"], t["hG"], t["hO"], off, pad, align, ke, mc)[0]
    return gI.double() + bbI.double() + tI.double()


def _run(t, off, pad, align, ke, mc, order, accumulate, with_cI=False):
    x = {k: v.to(DEV) for k, v in t.items()}
    o = off.to(DEV)
    sc = ops.StepContext(points_order=order, accumulate=accumulate)
    ops.forward(x["inp"], x["grid"], o, pad, align, ke, mc, ctx=sc)
    a = ops.backward(x["gOut"], x["inp"], x["grid"], o, pad, align, True, ke, mc, ctx=sc)
    b = ops.backward_backward(x["cI"] if with_cI else None, x["cG"], x["inp"], x["grid"], x["gOut"], o, pad, align, with_cI,
                              ke, mc, ctx=sc)
    c = ops.bbb_fused(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], x["hO"], o, pad, align, ke, mc, ctx=sc)
    torch.cuda.synchronize()
    return sc, a, b, c


@pytest.mark.parametrize("order", ["random", "coherent"])
@pytest.mark.parametrize("N,C,size,P", [(3, 16, (40, 33), 30011), (2, 5, (64, 64), 70001), (4, 32, (24, 24), 20000)])
def test_stages_add_into_one_accumulator(order, N, C, size, P):
    """every scatter stage of a step adds natively (no fallback) and the total equals the oracle's three gradients summed;
    the per-point outputs are the ones of the plain stages"""
    t = _case(N, C, size, P, seed=31 + C)
    off = offsets(N, True)
    ops.force_path(2)
    try:
        sc, a, b, c = _run(t, off, 0, True, 0, True, order, True)
        assert a[0] is None and b[0] is None and c[0] is None, "an accumulating context returns no grad_input"
        assert (sc.acc_native, sc.acc_fallback) == (3, 0)
        kind = sc._acc[0]
        assert kind == (_lib.ACC_CHANNELS_LAST if order == "coherent" else _lib.ACC_NCHW)
        total = sc.grad_input_sum()
        torch.cuda.synchronize()
        assert sc.grad_input_sum() is None, "the accumulator is handed over once"
        assert_close(total, _oracle_sum(t, off, 0, True, 0, True), "sum of the three input-shaped gradients")
        sc2, a2, b2, c2 = _run(t, off, 0, True, 0, True, order, False)
        assert_close(total, a2[0].double() + b2[0].double() + c2[0].double(), "accumulated vs stage by stage", 2e-6)
        for got, want, nm in ((a[1], a2[1], "grad_grid"), (b[1], b2[1], "bb grad_grid"), (b[2], b2[2], "bb grad_grad_out"),
                              (c[1], c2[1], "bbb grad_grad_out")):
            assert torch.equal(got, want), nm
    finally:
        ops.force_path(0)


def test_a_stage_on_the_other_path_falls_back_and_is_still_summed():
    """the second backward with grad_out_input leaves the coherent kernels (general path, NCHW sums) in a step whose
    accumulator is channels-last: it is run the plain way and added -- same total"""
    N, C, size, P = 3, 8, (32, 32), 40000
    t = _case(N, C, size, P, seed=77)
    off = offsets(N, True)
    ops.force_path(2)
    try:
        sc, a, b, c = _run(t, off, 0, True, 2, True, "coherent", True, with_cI=True)
        assert (sc.acc_native, sc.acc_fallback) == (2, 1)
        total = sc.grad_input_sum()
        torch.cuda.synchronize()
        assert_close(total, _oracle_sum(t, off, 0, True, 2, True, with_cI=True), "sum with a fallback stage")
    finally:
        ops.force_path(0)


def test_paths_without_native_accumulation_fall_back():
    """3D, and 2D beyond the fast path's channel count (channel groups): cs_accumulator_kind says none, the context sums"""
    t = _case(2, 4, (12, 12, 12), 20000, seed=5, d=3)
    off = offsets(2, True)
    ops.force_path(2)
    try:
        sc, a, b, c = _run(t, off, 0, True, 2, True, None, True)
        assert (sc.acc_native, sc.acc_fallback) == (0, 3)
        assert_close(sc.grad_input_sum(), _oracle_sum(t, off, 0, True, 2, True), "3D sum")
        t = _case(2, 40, (16, 16), 70000, seed=6)
        sc, a, b, c = _run(t, off, 0, True, 0, True, "random", True)
        assert sc.acc_native == 0 and sc.acc_fallback == 3
        assert_close(sc.grad_input_sum(), _oracle_sum(t, off, 0, True, 0, True), "channel groups sum")
    finally:
        ops.force_path(0)


def test_abi_refuses_a_mismatched_accumulator_without_writing():
    """CS_ERR_UNSUPPORTED for the other kind, with the accumulator untouched; CS_ERR_INVALID for an unknown kind"""
    N, C, H, P = 2, 16, 32, 70000
    t = {k: v.to(DEV) for k, v in _case(N, C, (H, H), P, seed=9).items()}
    off = offsets(N, True).to(DEV)
    lib = _lib.load()
    acc = torch.full((N, C, H, H), 3.0, device=DEV)
    gG = torch.empty_like(t["grid"])
    st = torch.cuda.current_stream().cuda_stream
    need = lib.cs_workspace_bytes(2, 1, N, C, 1, H, H, P, 0, 0, 0)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    for kind, flags, want in ((_lib.ACC_CHANNELS_LAST, 0, -2), (_lib.ACC_NCHW, _lib.POINTS_COHERENT, -2), (7, 0, -1)):
        lay = _lib.CotangentLayout(C * P, C * P, 0, 0, 0, kind, 0)
        rc = lib.cs2d_backward(t["gOut"].data_ptr(), t["inp"].data_ptr(), t["grid"].data_ptr(), off.data_ptr(), acc.data_ptr(),
                               gG.data_ptr(), N, C, H, H, P, 0, 1, flags, 1, lay, None, None, ws.data_ptr(), need, st)
        torch.cuda.synchronize()
        assert rc == want, (kind, flags, rc)
        assert float(acc.min()) == 3.0 and float(acc.max()) == 3.0
    assert lib.cs_accumulator_kind(2, N, C, 1, H, H, P, 0) == _lib.ACC_NCHW
    assert lib.cs_accumulator_kind(2, N, C, 1, H, H, P, _lib.POINTS_COHERENT) == _lib.ACC_CHANNELS_LAST
    assert lib.cs_accumulator_kind(3, N, C, H, H, H, P, 0) == _lib.ACC_NONE


def test_debug_knobs_cannot_make_the_shipped_library_wrong():
    """the ablation bits of cs_debug_coherent_tuning only exist in -DCS_COH_DEBUG builds: with them 'set', the default
    library still computes the right gradient"""
    N, C, size, P = 2, 16, (32, 32), 70000
    t = _case(N, C, size, P, seed=3)
    off = offsets(N, True)
    lib = _lib.load()
    lib.cs_debug_coherent_tuning(0, 7)
    try:
        x = {k: v.to(DEV) for k, v in t.items()}
        gI = ops.backward(x["gOut"], x["inp"], x["grid"], off.to(DEV), 0, True, True, 0, True,
                          ctx=ops.StepContext(points_order="coherent"))[0]
        want = cs_oracle.backward(t["gOut"], t["inp"], t["grid"], off, 0, True, True, 0, True)[0]
        assert_close(gI, want, "grad_input with ablation bits requested")
    finally:
        lib.cs_debug_coherent_tuning(0, 0)


def test_a_helmholtz_step_scatters_only_where_the_engine_asked():
    """BASELINE configs[2]'s call pattern through torch.autograd (reference test/test_2d.py:55-127, :221-240): the graph
    runs 11 sampler stages and only 5 of them produce an input-shaped gradient -- the derivatives taken with
    autograd.grad(u, x, create_graph=True) never use d/d cells, and the autograd layer asks the engine
    (functions._engine_wants, a private torch hook) before scattering.  If that question silently stops being answered,
    every stage scatters again (the reference's behaviour: correct, several ms slower): this count catches it."""
    from cosinesampler_amd import CosineSampler2d
    N, C, H, P = 4, 8, 32, 20000
    g = torch.Generator().manual_seed(5)
    cells = torch.rand(N, C, H, H, generator=g).to(DEV).requires_grad_(True)
    W1 = (torch.randn(16, C, generator=g) * 0.5).to(DEV)
    W2 = (torch.randn(1, 16, generator=g) * 0.5).to(DEV)
    x = (torch.rand(P, 1, generator=g) * 2 - 1).to(DEV).requires_grad_(True)
    y = (torch.rand(P, 1, generator=g) * 2 - 1).to(DEV).requires_grad_(True)
    ones = torch.ones(P, 1, device=DEV)
    ops.call_counts.clear()
    grid = torch.cat([x, y], -1).view(1, 1, P, 2).repeat(N, 1, 1, 1)
    feat = CosineSampler2d.apply(cells, grid, "zeros", True, "cosine", True).sum(0)
    u = torch.tanh(feat.view(C, -1).t() @ W1.t()) @ W2.t()
    u_x, u_y = torch.autograd.grad(u, (x, y), ones, create_graph=True)
    (u_xx,) = torch.autograd.grad(u_x, x, ones, create_graph=True)
    (u_yy,) = torch.autograd.grad(u_y, y, ones, create_graph=True)
    before = {k: list(v) for k, v in ops.call_counts.items()}
    assert sum(v[1] for v in before.values()) == 0, "taking u_x, u_xx scattered: %r" % before
    loss = torch.mean((u_xx + u_yy + 4.0 * u) ** 2)
    (gc,) = torch.autograd.grad(loss, cells)
    torch.cuda.synchronize()
    calls = sum(v[0] for v in ops.call_counts.values())
    scattered = sum(v[1] for v in ops.call_counts.values())
    assert (calls, scattered) == (11, 5), "sampler stages / of which scattering: %r" % ops.call_counts

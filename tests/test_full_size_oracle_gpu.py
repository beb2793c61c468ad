"""GPU (MI355X): EVERY output of every stage at BASELINE.json's full sizes against the CPU oracle, element by element --
the input-shaped gradients included (the tensors every scatter strategy exists for: reference 2d.cu:464-505, :661-712,
:850-888; 3d.cu:373-584, :587-870, :875-1071), on points as drawn and on points in cell order.

The oracle (oracle/cs_oracle.c, OpenMP over n) takes a few seconds per stage at these sizes on the box's host cores.
Tolerance: helpers.REL_TOL = 1e-5 relative per tensor (north_star: "all higher grads within 1e-5 fp32"), kernels and
oracle sharing the source index bit for bit.  The crowded reference shapes (96 tables of 16 x 16 cells, 1600 terms per
node) are compared with the oracle build that sums the gradients in double (cs_oracle.double_accumulation), so that the
checker's own serial fp32 rounding is not what the tolerance measures."""
import pytest
import torch

from cosinesampler_amd import CosineSampler2d, multicell_offset, ops
from helpers import assert_close, offsets, rel_err
from oracle import cs_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run_all(d, N, C, size, P, ke, order, seed, dacc=False, tol=1e-5):
    g = torch.Generator().manual_seed(seed)
    inp = torch.rand((N, C) + (size,) * d, generator=g)
    pts = torch.rand(P, d, generator=g) * 2 - 1
    if order == "ordered":
        pts = ops.sort_points(pts.to(DEV), (size,) * d)[0].cpu()
    grid = pts.view((1,) * d + (P, d)).repeat((N,) + (1,) * (d + 1)).contiguous()
    oshape = (N, C) + (1,) * (d - 1) + (P,)
    gOut, hO = torch.randn(oshape, generator=g), torch.randn(oshape, generator=g)
    cG, hG = torch.randn(grid.shape, generator=g), torch.randn(grid.shape, generator=g)
    off = offsets(N, True)
    x = [t.to(DEV) for t in (inp, grid, gOut, hO, cG, hG, off)]
    inp_d, grid_d, gOut_d, hO_d, cG_d, hG_d, off_d = x
    sc = ops.StepContext(points_order="coherent" if (order == "ordered" and d == 2) else "random")
    out = ops.forward(inp_d, grid_d, off_d, 0, True, ke, True, ctx=sc)
    gI, gG = ops.backward(gOut_d, inp_d, grid_d, off_d, 0, True, True, ke, True, ctx=sc)
    bbI, bbG, bbO = ops.backward_backward(None, cG_d, inp_d, grid_d, gOut_d, off_d, 0, True, False, ke, True, ctx=sc)
    tI, tO = ops.bbb_fused(inp_d, grid_d, gOut_d, cG_d, hG_d, hO_d, off_d, 0, True, ke, True, ctx=sc)
    torch.cuda.synchronize()
    got = dict(out=out, gI=gI, gG=gG, bbI=bbI, bbG=bbG, bbO=bbO, tI=tI, tO=tO)
    got = {k: v.cpu() for k, v in got.items()}
    del out, gI, gG, bbI, bbG, bbO, tI, tO, x, sc
    torch.cuda.empty_cache()

    def check():
        want = dict(out=cs_oracle.forward(inp, grid, off, 0, True, ke, True))
        assert_close(got.pop("out"), want.pop("out"), "forward", tol)
        w = cs_oracle.backward(gOut, inp, grid, off, 0, True, True, ke, True)
        assert_close(got.pop("gI"), w[0], "backward grad_input", tol)
        assert_close(got.pop("gG"), w[1], "backward grad_grid", tol)
        w = cs_oracle.backward_backward(None, cG, inp, grid, gOut, off, 0, True, False, ke, True)
        assert_close(got.pop("bbI"), w[0], "second backward grad_input", tol)
        assert_close(got.pop("bbG"), w[1], "second backward grad_grid", tol)
        assert_close(got.pop("bbO"), w[2], "second backward grad_grad_out", tol)
        w = cs_oracle.bbb_fused(inp, grid, gOut, cG, hG, hO, off, 0, True, ke, True)
        assert_close(got.pop("tI"), w[0], "third backward grad_input", tol)
        assert_close(got.pop("tO"), w[1], "third backward grad_grad_out", tol)

    if dacc:
        with cs_oracle.double_accumulation():
            check()
    else:
        check()


@pytest.mark.parametrize("order", ["drawn", "ordered"])
def test_configs1_every_output_matches_the_oracle(order):
    """BASELINE configs[1]: 2D cosine, multicell, N=16 C=16 256^2, P=2^20 -- tile walkers (drawn) / coherent kernels (ordered)"""
    _run_all(2, 16, 16, 256, 1 << 20, 0, order, seed=101)


@pytest.mark.parametrize("order", ["drawn", "ordered"])
def test_configs3_every_output_matches_the_oracle(order):
    """BASELINE configs[3]: 3D smooth-step, multicell, N=8 C=8 128^3, P=2^19 -- the tile path"""
    _run_all(3, 8, 8, 128, 1 << 19, 2, order, seed=103)


@pytest.mark.parametrize("d,N,C,size,P", [(2, 96, 4, 16, 100000), (3, 50, 4, 16, 100000)])
@pytest.mark.parametrize("order", ["drawn", "ordered"])
def test_reference_test_shapes_match_the_double_accumulating_oracle(d, N, C, size, P, order):
    """the reference's own test shapes (test/test_2d.py:26-38, test/test_3d.py:19-32): crowded tables, 1600 / 100 terms
    per node -- against the oracle that sums them in double"""
    _run_all(d, N, C, size, P, 0, order, seed=107, dacc=True)


def test_full_size_helmholtz_step_matches_the_oracle_backed_chain(monkeypatch):
    """BASELINE configs[2] at full size: u, u_x, u_xx, u_yy and d loss / d cells of the PIXEL-style Helmholtz step
    (reference test/test_2d.py:36-127, :221-240 pattern) through torch.autograd -- the product on the GPU against the SAME
    autograd chain run on the host with the CPU oracle in place of the kernels (tests/oracle_backend.py).  Third-order
    accuracy at the target config: 1e-5 relative on d loss / d cells (north_star)."""
    import oracle_backend
    N, C, H, P = 16, 16, 256, 1 << 20
    g = torch.Generator().manual_seed(131)
    cells0 = torch.rand(N, C, H, H, generator=g)
    W1 = torch.randn(16, C, generator=g) * 0.5
    W2 = torch.randn(1, 16, generator=g) * 0.5
    xy = torch.rand(P, 2, generator=g) * 2 - 1

    def step(dev):
        cells = cells0.to(dev).clone().requires_grad_(True)
        x = xy[:, :1].to(dev).clone().requires_grad_(True)
        y = xy[:, 1:].to(dev).clone().requires_grad_(True)
        ones = torch.ones(P, 1, device=dev)
        grid = torch.cat([x, y], -1).view(1, 1, P, 2).repeat(N, 1, 1, 1)
        feat = CosineSampler2d.apply(cells, grid, "zeros", True, "cosine", True).sum(0)
        u = torch.tanh(feat.view(C, -1).t() @ W1.to(dev).t()) @ W2.to(dev).t()
        u_x, u_y = torch.autograd.grad(u, (x, y), ones, create_graph=True)
        (u_xx,) = torch.autograd.grad(u_x, x, ones, create_graph=True)
        (u_yy,) = torch.autograd.grad(u_y, y, ones, create_graph=True)
        loss = torch.mean((u_xx + u_yy + 4.0 * u) ** 2)
        (gc,) = torch.autograd.grad(loss, cells)
        return {k: v.detach().cpu() for k, v in dict(u=u, u_x=u_x, u_y=u_y, u_xx=u_xx, u_yy=u_yy, gc=gc).items()}

    ops.points_order("random")
    try:
        got = step(DEV)
        torch.cuda.synchronize()
    finally:
        ops.points_order("auto")
    torch.cuda.empty_cache()
    oracle_backend.install(monkeypatch)
    want = step("cpu")
    for k in ("u", "u_x", "u_y", "u_xx", "u_yy", "gc"):
        assert_close(got[k], want[k], "Helmholtz step, %s" % k, 1e-5)


def test_streams_of_two_to_the_31_elements():
    """The case the reference instantiates its 64-bit-index kernels for (2d.cu:906-933: canUse32BitIndexMath fails once a
    tensor holds 2^31 elements): N=16 C=16 64^2 P=2^23 -- every channel-major stream has exactly 2^31 elements (8 GiB).
    forward + backward + fused third backward on the points as drawn (plan, records, walkers: 64-bit element offsets,
    32-bit sample ids) against the CPU oracle: the whole grad_input of both scatter stages and slices of every p-ordered
    output; then the same points in cell order (coherent kernels, 32-bit buffer offsets) must give the same grad_input."""
    N, C, H, P, K = 16, 16, 64, 1 << 23, 4096
    assert N * C * P == 1 << 31
    g = torch.Generator(device=DEV).manual_seed(1234)
    inp = torch.rand(N, C, H, H, device=DEV, generator=g)
    xy = torch.rand(P, 2, device=DEV, generator=g) * 2 - 1
    grid = xy.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
    gOut = torch.randn(N, C, 1, P, device=DEV, generator=g)
    hO = torch.randn(N, C, 1, P, device=DEV, generator=g)
    cG = torch.randn(N, 1, P, 2, device=DEV, generator=g)
    hG = torch.randn(N, 1, P, 2, device=DEV, generator=g)
    off = multicell_offset(N, True, DEV)
    sc = ops.StepContext(points_order="random")
    out = ops.forward(inp, grid, off, 0, True, 0, True, ctx=sc)
    gI, gG = ops.backward(gOut, inp, grid, off, 0, True, True, 0, True, ctx=sc)
    tI, tO = ops.bbb_fused(inp, grid, gOut, cG, hG, hO, off, 0, True, 0, True, ctx=sc)
    torch.cuda.synchronize()
    del sc
    # slices of the p-ordered outputs: the last table (the highest element offsets), points from the far end
    n, sl = N - 1, slice(P - K, P)
    c_inp, c_off = inp[n:n + 1].cpu(), off[n:n + 1].cpu()
    c_grid = grid[n:n + 1, :, sl].contiguous().cpu()
    ps = lambda t: t[n:n + 1, :, :, sl].contiguous().cpu()
    gs = lambda t: t[n:n + 1, :, sl].contiguous().cpu()
    assert_close(ps(out), cs_oracle.forward(c_inp, c_grid, c_off, 0, True, 0, True), "2^31: forward slice")
    assert_close(gs(gG), cs_oracle.backward(ps(gOut), c_inp, c_grid, c_off, 0, True, True, 0, True)[1], "2^31: grad_grid slice")
    assert_close(ps(tO), cs_oracle.bbb_fused(c_inp, c_grid, ps(gOut), gs(cG), gs(hG), ps(hO), c_off, 0, True, 0, True)[1],
                 "2^31: third-backward grad_grad_out slice")
    del out, tO, gG
    # the same points in cell order on the coherent kernels: grad_input does not depend on the order of the points
    xy_s, perm = ops.sort_points(xy, (H, H))
    grid_s = xy_s.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
    sc = ops.StepContext(points_order="coherent")
    gI_s, _ = ops.backward(gOut[..., perm].contiguous(), inp, grid_s, off, 0, True, True, 0, True, ctx=sc)
    torch.cuda.synchronize()
    assert_close(gI_s, gI, "2^31: coherent kernels, grad_input", 2e-6 * 8)      # 2^17 terms per node, two summation orders
    del gI_s, grid_s, sc, xy_s, perm
    torch.cuda.empty_cache()
    # the whole input-shaped gradients against the oracle (double accumulation: 2^17 terms per node)
    c = lambda t: t.cpu()
    with cs_oracle.double_accumulation():
        w_gI = cs_oracle.backward(c(gOut), c(inp), c(grid), c(off), 0, True, True, 0, True)[0]
        assert_close(gI, w_gI, "2^31: backward grad_input")
        del w_gI
        w_tI = cs_oracle.bbb_fused(c(inp), c(grid), c(gOut), c(cG), c(hG), c(hO), c(off), 0, True, 0, True)[0]
        assert_close(tI, w_tI, "2^31: third-backward grad_input")

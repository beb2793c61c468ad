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

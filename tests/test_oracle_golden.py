"""CPU: pin oracle/cs_oracle.c (the C restatement of the reference CUDA kernels) and
oracle/composite.py against the golden vectors captured from the reference's own ground truth
(tests/golden/make_golden.py), and against torch.nn.functional.grid_sample for the linear kernel."""
import pytest
import torch
import torch.nn.functional as F

from helpers import (KERNEL_ENUM, PAD, assert_close, axis_only, load, offsets, parse_stage_name, rel_err,
                     stage_fixtures)
from oracle import composite, cs_oracle


@pytest.mark.parametrize("name", stage_fixtures())
def test_c_oracle_stage_fixtures(name):
    d, kernel, mc = parse_stage_name(name)
    fx = load(name)
    cells, grid, gOut = fx["cells"], fx["grid"], fx["gOut"]
    off = offsets(cells.shape[0], mc)
    ke, pad, align = KERNEL_ENUM[kernel], PAD["zeros"], True

    out = cs_oracle.forward(cells, grid, off, pad, align, ke, mc)
    assert_close(out, fx["out"], name + " fwd out")

    gI, gG = cs_oracle.backward(gOut, cells, grid, off, pad, align, True, ke, mc)
    assert_close(gI, fx["gI"], name + " bwd grad_input")
    assert_close(gG, fx["gG"], name + " bwd grad_grid")
    none_gi, gG2 = cs_oracle.backward(gOut, cells, grid, off, pad, align, False, ke, mc)
    assert none_gi is None and torch.equal(gG, gG2)

    # general cotangents: gInput and ggOut are exact in both ops; gGrid only in 3D (App. B Q3)
    bbI, bbG, bbO = cs_oracle.backward_backward(fx["cI"], fx["cG"], cells, grid, gOut, off, pad, align, True, ke, mc)
    assert_close(bbI, fx["bbI"], name + " bb gInput")
    assert_close(bbO, fx["bbO"], name + " bb ggOut")
    if d == 3:
        assert_close(bbG, fx["bbG"], name + " bb gGrid (3D: mixed terms + gOutInput term)")

    for j in range(d):
        cGj = axis_only(fx["cG"], j)
        hGj = axis_only(fx["hG"], j)
        bI, bG, bO = cs_oracle.backward_backward(None, cGj, cells, grid, gOut, off, pad, align, False, ke, mc)
        assert_close(bI, fx["bbj%d_I" % j], name + " bb[j=%d] gInput" % j)
        assert_close(bO, fx["bbj%d_O" % j], name + " bb[j=%d] ggOut" % j)
        if d == 3:
            assert_close(bG, fx["bbj%d_G" % j], name + " bb[j=%d] gGrid" % j)
        else:  # 2D keeps the pure term only: compare the component along j against max of the exact tensor
            ref = fx["bbj%d_G" % j]
            den = max(float(ref.abs().max()), 1e-30)
            assert float((bG[..., j] - ref[..., j]).abs().max()) / den <= 1e-5, name + " bb[j] gGrid_j"
        tI, tO = cs_oracle.bbb_fused(cells, grid, gOut, cGj, hGj, fx["hO"], off, pad, align, ke, mc)
        assert_close(tI, fx["bbbj%d_I" % j], name + " bbb[j=%d] d/dinput" % j)
        assert_close(tO, fx["bbbj%d_O" % j], name + " bbb[j=%d] d/dgOut" % j)


@pytest.mark.parametrize("name", stage_fixtures())
def test_composite_matches_reference_ground_truth(name):
    d, kernel, mc = parse_stage_name(name)
    fx = load(name)
    cells = fx["cells"].clone().requires_grad_(True)
    grid = fx["grid"].clone().requires_grad_(True)
    gOut = fx["gOut"].clone().requires_grad_(True)
    out = composite.grid_sample_nd(cells, grid, kernel, mc, True)
    assert_close(out, fx["out"], name + " composite out")
    gI, gG = torch.autograd.grad(out, (cells, grid), gOut, create_graph=True)
    assert_close(gI, fx["gI"], name + " composite gI")
    assert_close(gG, fx["gG"], name + " composite gG")
    s = (gI * fx["cI"]).sum() + (gG * fx["cG"]).sum()
    bbI, bbG, bbO = torch.autograd.grad(s, (cells, grid, gOut), allow_unused=True)
    assert_close(bbI, fx["bbI"], name + " composite bbI")
    assert_close(bbO, fx["bbO"], name + " composite bbO")
    if bbG is not None:
        assert_close(bbG, fx["bbG"], name + " composite bbG", tol=2e-5)


@pytest.mark.parametrize("pad", ["zeros", "border"])
@pytest.mark.parametrize("shape", [(1, 1, 32, 32, 1024), (3, 5, 9, 14, 333)])
def test_linear_matches_torch_grid_sample_2d(pad, shape):
    """README.md:26-27 of the reference: bilinear + multicell=False == torch grid_sample
    (BASELINE.json configs[0]).  Includes out-of-range points to exercise the padding."""
    N, C, H, W, P = shape
    g = torch.Generator().manual_seed(3)
    inp = torch.rand(N, C, H, W, generator=g, requires_grad=True)
    grid = (torch.rand(N, 1, P, 2, generator=g) * 2.6 - 1.3).requires_grad_(True)
    gOut = torch.randn(N, C, 1, P, generator=g)
    ref = F.grid_sample(inp, grid, mode="bilinear", padding_mode=pad, align_corners=True)
    rI, rG = torch.autograd.grad(ref, (inp, grid), gOut)
    off = offsets(N, False)
    out = cs_oracle.forward(inp.detach(), grid.detach(), off, PAD[pad], True, 1, False)
    gI, gG = cs_oracle.backward(gOut, inp.detach(), grid.detach(), off, PAD[pad], True, True, 1, False)
    assert rel_err(out, ref.detach()) <= 2e-6
    assert rel_err(gI, rI) <= 1e-5
    assert rel_err(gG, rG) <= 1e-5


@pytest.mark.parametrize("pad", ["zeros", "border"])
@pytest.mark.parametrize("align", [True, False])
def test_linear_matches_torch_grid_sample_3d(pad, align):
    N, C, D, H, W, P = 2, 3, 5, 7, 6, 301
    g = torch.Generator().manual_seed(4)
    inp = torch.rand(N, C, D, H, W, generator=g, requires_grad=True)
    grid = (torch.rand(N, 1, 1, P, 3, generator=g) * 2.6 - 1.3).requires_grad_(True)
    gOut = torch.randn(N, C, 1, 1, P, generator=g)
    ref = F.grid_sample(inp, grid, mode="bilinear", padding_mode=pad, align_corners=align)
    rI, rG = torch.autograd.grad(ref, (inp, grid), gOut)
    off = offsets(N, False)
    out = cs_oracle.forward(inp.detach(), grid.detach(), off, PAD[pad], align, 1, False)
    gI, gG = cs_oracle.backward(gOut, inp.detach(), grid.detach(), off, PAD[pad], align, True, 1, False)
    assert rel_err(out, ref.detach()) <= 2e-6
    assert rel_err(gI, rI) <= 1e-5
    assert rel_err(gG, rG) <= 1e-5


def test_2d_forward_ignores_align_corners_quirk():
    """Reference 2d.cu:307-308 hard-wires align_corners=1 in the 2D forward (App. B Q1)."""
    g = torch.Generator().manual_seed(5)
    inp = torch.rand(2, 2, 8, 8, generator=g)
    grid = torch.rand(2, 1, 50, 2, generator=g) * 2 - 1
    off = offsets(2, True)
    a = cs_oracle.forward(inp, grid, off, 0, True, 0, True)
    b = cs_oracle.forward(inp, grid, off, 0, False, 0, True)
    assert torch.equal(a, b)


def test_oracle_is_clean_under_asan_and_ubsan():
    """the CPU restatement under ASan / UBSan -- `make -C oracle asan-run` compiles cs_oracle.c into
    oracle/asan_driver.c with -fsanitize=address,undefined and runs every function over exactly-sized heap buffers (all
    padding modes, both align_corners settings, the three kernels, points far outside the range, optional tensors absent)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "asan-run"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "no finding" in r.stdout

#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own ground truth.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

The reference ships no golden files; its tests compare the CUDA op with a pure-PyTorch
composite, test/grid_sampler.py, differentiated by autograd (reference test/test_2d.py:137-244,
test/test_3d.py).  This script imports THAT module from where it lies (never copied into the
repo, bytecode writing disabled), evaluates it on small seeded inputs and stores inputs +
outputs as data.  grid_sampler.py hard-codes `.to("cuda")` for the multicell offset
(grid_sampler.py:34, :121); there is no GPU here, so a harness-side shim makes
`Tensor.to("cuda")` a no-op while the reference code runs.

What is stored (all fp32):
  stage_{2d,3d}_{kernel}_{mc|nomc}.npz   one op stage at a time, cotangents seeded:
     fwd   out
     bwd   (gI, gG)        = grad(<out, gOut>, (cells, grid))
     bb    (bbI, bbG, bbO) = grad(<gI, cI> + <gG, cG>, (cells, grid, gOut))        general cotangents
     bbj{j}  the same with cI = 0 and cG non-zero on axis j only                   (see below)
     bbbj{j} (tI, tO) = grad(<bbG_j, hG_j> + <bbO_j, hO>, (cells, gOut)), hG_j on axis j only
  pixel_{2d,3d}.npz   the PIXEL-style pipeline of the reference tests: sampler -> sum over n ->
     MLP -> u, u_x.., u_xx.., d/dcells of each, and d loss / d cells.

Why axis-aligned cotangents: the reference op is not the exact derivative everywhere
(SURVEY.md App. B Q3-Q5): the 2D second backward drops mixed second derivatives and the
cI -> gGrid term, and both third backwards keep pure second derivatives only.  All of those
vanish for axis-aligned cG/hG with cI = 0 -- which is exactly what grad(u_x, x) and
grad(u_xx, cells) produce.
"""
import contextlib
import importlib.util
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/test/grid_sampler.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_grid_sampler", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@contextlib.contextmanager
def cuda_is_cpu():
    orig = torch.Tensor.to

    def to(self, *a, **k):
        if (a and a[0] == "cuda") or k.get("device") == "cuda":
            return self
        return orig(self, *a, **k)

    torch.Tensor.to = to
    try:
        yield
    finally:
        torch.Tensor.to = orig


REF_NAME = {  # API kernel string -> reference-oracle `step` string
    "cosine": "cosine", "smooth-step": "smoothstep", "bilinear": "bilinear", "trilinear": "trilinear"}


def ref_sample(ref, cells, grid, kernel, multicell):
    with cuda_is_cpu():
        if cells.dim() == 4:
            return ref.grid_sample_2d(cells, grid, step=REF_NAME[kernel], offset=multicell)
        N, C = cells.shape[:2]
        out = ref.grid_sample_3d(cells, grid, step=REF_NAME[kernel], offset=multicell)  # (N,C,1,P)
        return out.view(N, C, 1, 1, -1)


def make_points(P, d, gen, multicell):
    xy = torch.rand(P, d, generator=gen) * 2 - 1
    # A few special points: exact corners / centre / a cell boundary.  Without multicell g = +1
    # lands on the last node, whose right neighbour is out of range: the reference oracle CLAMPS
    # that index (grid_sampler.py:68-72) while the op ZERO-PADS it (2d.cu:342-353), so their
    # derivatives differ there by design -- keep those points just inside.
    top = 1.0 if multicell else 0.999
    xy[0] = -1.0
    xy[1] = top
    xy[2] = 0.0
    xy[3, 0] = top
    xy[4, -1] = -1.0
    return xy


def g0(t, like):
    return torch.zeros_like(like) if t is None else t


def stage_case(ref, d, kernel, multicell, N, C, S, P, seed):
    gen = torch.Generator().manual_seed(seed)
    cells = torch.rand((N, C) + (S,) * d, generator=gen).requires_grad_(True)
    pts = torch.stack([make_points(P, d, gen, multicell) for _ in range(N)])           # independent per n
    grid = pts.view((N,) + (1,) * (d - 1) + (P, d)).clone().requires_grad_(True)
    oshape = (N, C) + (1,) * (d - 1) + (P,)
    gOut = torch.randn(oshape, generator=gen).requires_grad_(True)
    cI = torch.randn(cells.shape, generator=gen)
    cG = torch.randn(grid.shape, generator=gen)
    hO = torch.randn(oshape, generator=gen)
    hG = torch.randn(grid.shape, generator=gen)

    out = ref_sample(ref, cells, grid, kernel, multicell)
    gI, gG = torch.autograd.grad(out, (cells, grid), gOut, create_graph=True)
    s = (gI * cI).sum() + (gG * cG).sum()
    bbI, bbG, bbO = torch.autograd.grad(s, (cells, grid, gOut), retain_graph=True, allow_unused=True)
    fx = dict(cells=cells, grid=grid, gOut=gOut, cI=cI, cG=cG, hO=hO, hG=hG,
              out=out, gI=gI, gG=gG, bbI=g0(bbI, cells), bbG=g0(bbG, grid), bbO=g0(bbO, gOut))
    for j in range(d):
        cGj = torch.zeros_like(cG)
        cGj[..., j] = cG[..., j]
        hGj = torch.zeros_like(hG)
        hGj[..., j] = hG[..., j]
        sj = (gG * cGj).sum()
        bI, bG, bO = torch.autograd.grad(sj, (cells, grid, gOut), create_graph=True, allow_unused=True)
        fx["bbj%d_I" % j], fx["bbj%d_G" % j], fx["bbj%d_O" % j] = g0(bI, cells), g0(bG, grid), g0(bO, gOut)
        s3 = 0
        if bG is not None and bG.requires_grad:
            s3 = s3 + (bG * hGj).sum()
        if bO is not None and bO.requires_grad:
            s3 = s3 + (bO * hO).sum()
        tI, tO = torch.autograd.grad(s3, (cells, gOut), retain_graph=True, allow_unused=True)
        fx["bbbj%d_I" % j], fx["bbbj%d_O" % j] = g0(tI, cells), g0(tO, gOut)
    meta = dict(d=d, kernel=kernel, multicell=multicell)
    return {k: v.detach().numpy().astype(np.float32) for k, v in fx.items()}, meta


def pixel_case(ref, d, N, C, S, P, seed):
    """Reference test/test_2d.py:26-240 (2D: Allen-Cahn-like residual) and test/test_3d.py:19-292
    (3D: u_xx+u_yy+u_zz+u), with the MLP weights stored in the fixture."""
    gen = torch.Generator().manual_seed(seed)
    kernel, multicell = "cosine", True
    cells = torch.rand((N, C) + (S,) * d, generator=gen).requires_grad_(True)
    coords = [(torch.rand(P, 1, generator=gen) * 2 - 1).requires_grad_(True) for _ in range(d)]
    W1 = torch.randn(16, C, generator=gen) * 0.5
    b1 = torch.randn(16, generator=gen) * 0.1
    W2 = torch.randn(1, 16, generator=gen) * 0.5
    b2 = torch.randn(1, generator=gen) * 0.1

    grid = torch.cat(coords, -1).view((1,) * d + (P, d)).repeat((N,) + (1,) * (d + 1))
    val = ref_sample(ref, cells, grid, kernel, multicell)
    feat = val.sum(0).view(C, -1).t()
    u = torch.tanh(feat @ W1.t() + b1) @ W2.t() + b2

    def grad(y, x):
        return torch.autograd.grad(y, x, torch.ones_like(y), retain_graph=True, create_graph=True)[0]

    fx = dict(cells=cells, W1=W1, b1=b1, W2=W2, b2=b2, u=u, u_cell=grad(u, cells))
    names = "xyz"[:d]
    first, second = [], []
    for j, nm in enumerate(names):
        fx["p_" + nm] = coords[j]
        uj = grad(u, coords[j])
        ujj = grad(uj, coords[j])
        first.append(uj)
        second.append(ujj)
        fx["u_" + nm], fx["u_" + nm * 2] = uj, ujj
        fx["u_%s_cell" % nm], fx["u_%s_cell" % (nm * 2)] = grad(uj, cells), grad(ujj, cells)
    if d == 2:
        f = first[1] * 2 + 5 * (u ** 3) - 5 * u - 0.0001 * second[0]   # test_2d.py:221
    else:
        f = second[0] + second[1] + second[2] + u                        # test_3d.py:270
    loss = torch.mean(f ** 2)
    fx["dloss"] = torch.autograd.grad(loss, cells)[0]
    return {k: v.detach().numpy().astype(np.float32) for k, v in fx.items()}


def main():
    ref = load_reference()
    os.makedirs(HERE, exist_ok=True)
    written = []
    for d, kernels, N, C, S, P in ((2, ("cosine", "smooth-step", "bilinear"), 4, 3, 16, 257),
                                   (3, ("cosine", "smooth-step", "trilinear"), 3, 2, 8, 257)):
        for kernel in kernels:
            for multicell in (True, False):
                fx, meta = stage_case(ref, d, kernel, multicell, N, C, S, P, seed=1000 * d + 7)
                name = "stage_%dd_%s_%s.npz" % (d, kernel.replace("-", ""), "mc" if multicell else "nomc")
                np.savez_compressed(os.path.join(HERE, name), **fx)
                written.append(name)
    # BASELINE.json configs[0]: 2D linear, multicell=False, N=1 C=1 H=W=32 P=1024
    fx, _ = stage_case(ref, 2, "bilinear", False, 1, 1, 32, 1024, seed=11)
    np.savez_compressed(os.path.join(HERE, "stage_2d_config1.npz"), **fx)
    written.append("stage_2d_config1.npz")
    np.savez_compressed(os.path.join(HERE, "pixel_2d.npz"), **pixel_case(ref, 2, 6, 4, 16, 300, seed=51))
    np.savez_compressed(os.path.join(HERE, "pixel_3d.npz"), **pixel_case(ref, 3, 5, 4, 8, 300, seed=6))
    written += ["pixel_2d.npz", "pixel_3d.npz"]
    total = sum(os.path.getsize(os.path.join(HERE, n)) for n in written)
    print("wrote %d fixtures, %.1f KiB" % (len(written), total / 1024))


if __name__ == "__main__":
    main()

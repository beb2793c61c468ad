"""Pure-PyTorch differentiable composite of the sampler -- TEST INFRASTRUCTURE ONLY.

Same role as the reference's own ground truth (reference test/grid_sampler.py): floor +
gather + blending weights, with every derivative order supplied by PyTorch autograd.  It is
this repo's own restatement (any dimensionality, true zero padding instead of index clamping,
any device) and is used
  * by tests/ as a second, exact-derivative checker next to the C restatement, and
  * by bench.py as the `cpu_baseline` leg (the "PyTorch-autograd CPU path").
It is validated against the imported reference oracle in tests/golden/make_golden.py.
Nothing under cosinesampler_amd/ imports it.

Math: SURVEY.md Appendix A.  Coordinates: reference 2d.cu:54-66 (unnormalize with the
multicell `size-1` and `+offset[n]`), kernels 2d.cu:239-261.
"""
import math

import torch

_KERNELS = {
    "cosine": lambda t: 0.5 * (1 - torch.cos(math.pi * t)),
    "linear": lambda t: t,
    "bilinear": lambda t: t,
    "trilinear": lambda t: t,
    "smooth-step": lambda t: t * t * (3 - 2 * t),
    "smoothstep": lambda t: t * t * (3 - 2 * t),
}


def multicell_offset(N, multicell, device=None):
    """offset[n] exactly as the reference builds it (mod2d.py:24-27)."""
    if multicell:
        return torch.linspace(0, 1 - (1 / N), N).to(device)
    return torch.zeros(N).to(device)


def grid_sample_nd(input, grid, kernel="cosine", multicell=True, align_corners=True):
    """input (N,C,*spatial), grid (N,*out,d) with grid[...,0] <-> last spatial axis.
    Zero padding.  Returns (N,C,*out).  Differentiable to any order in input and grid."""
    N, C = input.shape[:2]
    sp = list(input.shape[2:])
    d = len(sp)
    assert grid.shape[-1] == d and grid.shape[0] == N
    out_shape = tuple(grid.shape[1:-1])
    k = _KERNELS[kernel]
    off = multicell_offset(N, multicell, input.device).to(input.dtype).view(N, 1)
    g = grid.reshape(N, -1, d)
    P = g.shape[1]
    flat = input.reshape(N, C, -1)

    lows, w_lo, w_hi = [], [], []
    for j in range(d):  # j = 0 is x (last spatial axis)
        size = sp[d - 1 - j]
        if align_corners:
            i = ((g[..., j] + 1) / 2) * (size - 1 - (1 if multicell else 0)) + off
        else:
            i = ((g[..., j] + 1) * size - 1) / 2 + off
        lo = torch.floor(i.detach())
        t = (lo + 1) - i  # distance to the high corner; k(t) weights the LOW corner
        kl = k(t)
        lows.append(lo.long())
        w_lo.append(kl)
        w_hi.append(1 - kl)

    out = 0
    for a in range(1 << d):
        w = 1
        lin = torch.zeros(N, P, dtype=torch.long, device=input.device)
        ok = torch.ones(N, P, dtype=torch.bool, device=input.device)
        stride = 1
        for j in range(d):
            hi = (a >> j) & 1
            size = sp[d - 1 - j]
            idx = lows[j] + hi
            ok = ok & (idx >= 0) & (idx < size)
            lin = lin + idx.clamp(0, size - 1) * stride
            stride *= size
            w = w * (w_hi[j] if hi else w_lo[j])
        vals = torch.gather(flat, 2, lin.view(N, 1, P).expand(N, C, P))
        out = out + vals * (w * ok.to(input.dtype)).view(N, 1, P)
    return out.view((N, C) + out_shape)

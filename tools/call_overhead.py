#!/usr/bin/env python3
"""Host-side cost of one stage call (ctypes + torch allocations), measured with a tiny problem so that the GPU is
never the bottleneck: wall-clock microseconds per call, queue kept non-empty.  python tools/call_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosinesampler_amd import CosineSampler2d, multicell_offset, ops

dev = torch.device("cuda", 0)
N, C, H, P = 4, 4, 16, 1024
cells = torch.rand(N, C, H, H, device=dev)
grid = (torch.rand(N, 1, P, 2, device=dev) * 2 - 1)
gO = torch.randn(N, C, 1, P, device=dev)
cG = torch.randn(N, 1, P, 2, device=dev)
off = multicell_offset(N, True, dev)
sc = ops.StepContext()


def t(fn, reps=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / reps * 1e6


print("torch.empty_like + add (reference point)   %.1f us" % t(lambda: torch.add(cells, cells)))
print("ops.forward                                %.1f us" % t(lambda: ops.forward(cells, grid, off, 0, True, 0, True, ctx=sc)))
print("ops.backward (grad_input)                  %.1f us" % t(lambda: ops.backward(gO, cells, grid, off, 0, True, True, 0, True, ctx=sc)))
print("ops.backward_backward                      %.1f us" % t(lambda: ops.backward_backward(None, cG, cells, grid, gO, off, 0, True, False, 0, True, ctx=sc)))
cr = cells.clone().requires_grad_(True)
print("CosineSampler2d.apply (forward, autograd)  %.1f us" % t(lambda: CosineSampler2d.apply(cr, grid, "zeros", True, "cosine", True)))

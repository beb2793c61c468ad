#!/usr/bin/env python3
"""Do two half-size stages on two HIP streams overlap?  (decides whether splitting a stage into halves whose
tile kernel runs next to the other half's point kernel can pay)   python tools/overlap_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosinesampler_amd import multicell_offset, ops

dev = torch.device("cuda", 0)
C, H, P = 16, 256, 1 << 20


def make(N, seed):
    torch.manual_seed(seed)
    cells = torch.rand(N, C, H, H, device=dev)
    xy = torch.rand(P, 2, device=dev) * 2 - 1
    grid = xy.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
    return dict(cells=cells, grid=grid, gOut=torch.randn(N, C, 1, P, device=dev), cG=torch.randn(N, 1, P, 2, device=dev),
                hG=torch.randn(N, 1, P, 2, device=dev), hO=torch.randn(N, C, 1, P, device=dev),
                off=multicell_offset(16, True, dev)[:N].contiguous(), sc=ops.StepContext())


def bb(t):
    return ops.backward_backward(None, t["cG"], t["cells"], t["grid"], t["gOut"], t["off"], 0, True, False, 0, True, ctx=t["sc"])


def bbb(t):
    return ops.bbb_fused(t["cells"], t["grid"], t["gOut"], t["cG"], t["hG"], t["hO"], t["off"], 0, True, 0, True, ctx=t["sc"])


def timeit(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


full = make(16, 0)
a, b = make(8, 1), make(8, 2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, st in (("backward_backward", bb), ("bbb_fused", bbb)):
    t_full = timeit(lambda: st(full))
    t_seq = timeit(lambda: (st(a), st(b)))

    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ra = st(a)
        with torch.cuda.stream(s2):
            rb = st(b)
        cur.wait_stream(s1); cur.wait_stream(s2)
        return ra, rb
    t_par = timeit(both)
    print("%-18s N=16 one call %.3f ms | two N=8 calls in sequence %.3f | on two streams %.3f" % (name, t_full, t_seq, t_par), flush=True)

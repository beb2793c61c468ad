#!/usr/bin/env python3
"""Experiment: build the point plan on a second stream while the forward runs (it depends on the grid alone).
    python tools/overlap_plan.py   -> ms per step: plan in line / plan on a side stream (before / after the forward launch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosinesampler_amd import multicell_offset, ops

dev = torch.device("cuda", 0)
N, C, H, P = 16, 16, 256, 1 << 20
torch.manual_seed(0)
cells = torch.rand(N, C, H, H, device=dev)
xy = torch.rand(P, 2, device=dev) * 2 - 1
grid = xy.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
gOut = torch.randn(N, C, 1, P, device=dev); hO = torch.randn(N, C, 1, P, device=dev)
cG = torch.randn(N, 1, P, 2, device=dev); hG = torch.randn(N, 1, P, 2, device=dev)
off = multicell_offset(N, True, dev)
side = torch.cuda.Stream(priority=-1)


def step(mode):
    sc = ops.StepContext()
    main = torch.cuda.current_stream()
    if mode == 1:      # plan first, on the side stream; forward on the main one
        side.wait_stream(main)
        with torch.cuda.stream(side):
            sc.prepare_plan(cells, grid, off, 0, True, True)
    ops.forward(cells, grid, off, 0, True, 0, True, ctx=sc)
    if mode == 2:      # forward first, then the plan on the side stream
        with torch.cuda.stream(side):
            sc.prepare_plan(cells, grid, off, 0, True, True)
    if mode:
        main.wait_stream(side)
    else:
        sc.prepare_plan(cells, grid, off, 0, True, True)
    ops.backward(gOut, cells, grid, off, 0, True, True, 0, True, ctx=sc)
    ops.backward_backward(None, cG, cells, grid, gOut, off, 0, True, False, 0, True, ctx=sc)
    ops.bbb_fused(cells, grid, gOut, cG, hG, hO, off, 0, True, 0, True, ctx=sc)
    return sc


for mode in (0, 1, 2, 0, 1, 2):
    for _ in range(3):
        keep = step(mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        keep = step(mode)
    e1.record()
    torch.cuda.synchronize()
    print("mode %d: %.3f ms per step" % (mode, e0.elapsed_time(e1) / 10), flush=True)

#!/usr/bin/env python3
"""Experiment: BASELINE configs[3] with the point plan built on a second stream while the forward (and the table pack) run.
    python tools/overlap_plan3d.py   -> ms per step: plan in line (inside the first backward) / on a side stream"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosinesampler_amd import multicell_offset, ops

dev = torch.device("cuda", 0)
N, C, S, P = 8, 8, 128, 1 << 19
g = torch.Generator(device="cpu").manual_seed(11)
cells = torch.rand((N, C, S, S, S), generator=g).to(dev)
pts = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
grid = pts.view(1, 1, 1, P, 3).repeat(N, 1, 1, 1, 1).contiguous()
gOut = torch.randn((N, C, 1, 1, P), generator=g).to(dev); hO = torch.randn((N, C, 1, 1, P), generator=g).to(dev)
cG = torch.randn(grid.shape, generator=g).to(dev); hG = torch.randn(grid.shape, generator=g).to(dev)
off = multicell_offset(N, True, dev)
side = torch.cuda.Stream(priority=-1)


def step(mode):
    sc = ops.StepContext()
    main = torch.cuda.current_stream()
    if mode == 1:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            sc.prepare_plan(cells, grid, off, 0, True, True)
    ops.forward(cells, grid, off, 0, True, 2, True, ctx=sc)
    if mode == 2:
        with torch.cuda.stream(side):
            sc.prepare_plan(cells, grid, off, 0, True, True)
    if mode:
        main.wait_stream(side)
    ops.backward(gOut, cells, grid, off, 0, True, True, 2, True, ctx=sc)
    ops.backward_backward(None, cG, cells, grid, gOut, off, 0, True, False, 2, True, ctx=sc)
    ops.bbb_fused(cells, grid, gOut, cG, hG, hO, off, 0, True, 2, True, ctx=sc)
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
    print("3D mode %d: %.3f ms per step" % (mode, e0.elapsed_time(e1) / 10), flush=True)

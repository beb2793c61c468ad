#!/usr/bin/env python3
"""Development helper: the ordered-points step of bench.py (BASELINE configs[1] on points in cell order, accumulating
context, fresh StepContext per step) with per-stage HIP-event spans, for sweeps that need no rebuild:
    COSINESAMPLER_DEBUG=1 CS_CHUNKS=4,8,6,6 python tools/ordered_step.py [steps]
CS_CHUNKS = samples per wave of forward, backward, backward_backward, bbb_fused in batches of 64 (cs_debug_coherent_tuning)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosinesampler_amd import multicell_offset, ops

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
N, C, H, P = 16, 16, 256, 1 << 20
g = torch.Generator(device="cpu").manual_seed(0)
cells = torch.rand(N, C, H, H, generator=g).to(dev)
xy = (torch.rand(P, 2, generator=torch.Generator(device="cpu").manual_seed(1000)) * 2 - 1).to(dev)
xy, _ = ops.sort_points(xy, (H, H))
grid = xy.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
torch.manual_seed(1)
gOut = torch.randn(N, C, 1, P, device=dev); hO = torch.randn(N, C, 1, P, device=dev)
cG = torch.randn(N, 1, P, 2, device=dev); hG = torch.randn(N, 1, P, 2, device=dev)
off = multicell_offset(N, True, dev)
if os.environ.get("CS_CHUNKS") or os.environ.get("CS_WPB") or os.environ.get("CS_ABLATE"):
    f, b, bb, bbb = (int(x) for x in os.environ.get("CS_CHUNKS", "4,6,6,6").split(","))
    wpb = int(os.environ.get("CS_WPB", "1"))           # waves per workgroup (1..4)
    # CS_ABLATE: parts of the kernels switched off (results wrong; only in -DCS_COH_DEBUG builds of cs_coherent): 1 no
    # scatter-reduce, 2 no accumulator-window flush, 4 no products, 1024 no table-window loads, 2048 no scatter operands to LDS
    abl = int(os.environ.get("CS_ABLATE", "0"))
    assert ops._lib.load().cs_debug_coherent_tuning(f | b << 8 | bb << 16 | bbb << 24, wpb << 4 | abl) == 1, "needs COSINESAMPLER_DEBUG=1"


def step(ev=None):
    sc = ops.StepContext(points_order="coherent", accumulate=True)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if ev is not None else None
    if e: e[0].record()
    ops.forward(cells, grid, off, 0, True, 0, True, ctx=sc)
    if e: e[1].record()
    ops.backward(gOut, cells, grid, off, 0, True, True, 0, True, ctx=sc)
    if e: e[2].record()
    ops.backward_backward(None, cG, cells, grid, gOut, off, 0, True, False, 0, True, ctx=sc)
    if e: e[3].record()
    ops.bbb_fused(cells, grid, gOut, cG, hG, hO, off, 0, True, 0, True, ctx=sc)
    total = sc.grad_input_sum()
    if e:
        e[4].record()
        ev.append(e)
    return total


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
ev = []
for _ in range(steps):
    step(ev)
torch.cuda.synchronize()
st = [sum(e[i].elapsed_time(e[i + 1]) for e in ev) / len(ev) for i in range(4)]
print("ablate %s wpb %s chunks %-12s step %.3f ms | forward %.3f  backward %.3f  backward_backward %.3f  bbb_fused %.3f"
      % (os.environ.get("CS_ABLATE", "0"), os.environ.get("CS_WPB", "1"), os.environ.get("CS_CHUNKS", "default"), ms, *st), flush=True)

#!/usr/bin/env python3
"""Development helper: the bench's stage pipeline at config 2 plus the calls that do not want grad_input, fresh
StepContext per step (cold, as in bench.py).  Run under rocprofv3 --kernel-trace --stats for per-kernel times:
    CS_FORCE=4 python tools/pipe.py   (round-1 fat-row path)      CS_FORCE=2 python tools/pipe.py   (sorted path)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosinesampler_amd import multicell_offset, ops

dev = torch.device("cuda", 0)
N, C, H, P = 16, int(os.environ.get("CS_C", "16")), 256, 1 << 20
torch.manual_seed(0)
cells = torch.rand(N, C, H, H, device=dev)
xy = torch.rand(P, 2, device=dev) * 2 - 1
grid = xy.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
gOut = torch.randn(N, C, 1, P, device=dev); hO = torch.randn(N, C, 1, P, device=dev)
cG = torch.randn(N, 1, P, 2, device=dev); hG = torch.randn(N, 1, P, 2, device=dev)
off = multicell_offset(N, True, dev)
SD = {"": None, "bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("CS_STREAM", "")]
if SD is not None:
    gOut, hO = gOut.to(SD), hO.to(SD)
if os.environ.get("CS_FORCE"):
    ops.force_path(int(os.environ["CS_FORCE"]))
extra = os.environ.get("CS_EXTRA", "1") == "1"
for step in range(int(os.environ.get("CS_STEPS", "5"))):
    sc = ops.StepContext()
    ops.forward(cells, grid, off, 0, True, 0, True, ctx=sc, out_dtype=SD)
    ops.backward(gOut, cells, grid, off, 0, True, True, 0, True, ctx=sc)
    if extra:
        ops.backward(gOut, cells, grid, off, 0, True, False, 0, True, ctx=sc)
    ops.backward_backward(None, cG, cells, grid, gOut, off, 0, True, False, 0, True, ctx=sc)
    if extra:
        ops.backward_backward(None, cG, cells, grid, gOut, off, 0, True, False, 0, True, ctx=sc, want_grad_input=False)
    ops.bbb_fused(cells, grid, gOut, cG, hG, hO, off, 0, True, 0, True, ctx=sc)
torch.cuda.synchronize()

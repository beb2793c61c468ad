#!/usr/bin/env python3
"""Development helper: the four stages of the summed op (ops.*_sum_n, CS_SUM_OVER_N) at the headline shapes on ordered points,
next to the plain op on the same (repeated / expanded) inputs.  CS_ABLATE as tools/stage_time.py (4096: no window prefetch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosinesampler_amd import multicell_offset, ops

dev = torch.device("cuda", 0)
N, C, H, P = 16, int(os.environ.get("CS_C", "16")), 256, 1 << 20
torch.manual_seed(0)
cells = torch.rand(N, C, H, H, device=dev)
xy, _ = ops.sort_points(torch.rand(P, 2, device=dev) * 2 - 1, (H, H))
g1 = xy.view(1, 1, P, 2).contiguous()
gO = torch.randn(1, C, 1, P, device=dev); hO = torch.randn(1, C, 1, P, device=dev)
cG = torch.randn(1, 1, P, 2, device=dev); hG = torch.randn(1, 1, P, 2, device=dev)
off = multicell_offset(N, True, dev)
ops.points_order("coherent")
if os.environ.get("CS_ABLATE"):
    ops._lib.load().cs_debug_coherent_tuning(0, int(os.environ["CS_ABLATE"]))
sc = ops.StepContext()
gOe, hOe = gO.expand(N, -1, -1, -1), hO.expand(N, -1, -1, -1)
stages = [
    ("forward_sum_n", lambda: ops.forward_sum_n(cells, g1, off, 0, True, 0, True, ctx=sc)),
    ("  plain + sum", lambda: ops.forward(cells, g1, off, 0, True, 0, True, ctx=sc).sum(0, keepdim=True)),
    ("backward_sum_n", lambda: ops.backward_sum_n(gO, cells, g1, off, 0, True, True, 0, True, ctx=sc)),
    ("  plain", lambda: ops.backward(gOe, cells, g1, off, 0, True, True, 0, True, ctx=sc)),
    ("backward_sum_n (no grad_input)", lambda: ops.backward_sum_n(gO, cells, g1, off, 0, True, False, 0, True, ctx=sc)),
    ("  plain", lambda: ops.backward(gOe, cells, g1, off, 0, True, False, 0, True, ctx=sc)),
    ("backward_backward_sum_n", lambda: ops.backward_backward_sum_n(cG, cells, g1, gO, off, 0, True, 0, True, ctx=sc)),
    ("  plain + sum", lambda: ops.backward_backward(None, cG, cells, g1, gOe, off, 0, True, False, 0, True, ctx=sc)[2].sum(0, keepdim=True)),
    ("backward_backward_sum_n (no grad_input)", lambda: ops.backward_backward_sum_n(cG, cells, g1, gO, off, 0, True, 0, True, ctx=sc, want_grad_input=False)),
    ("bbb_fused_sum_n", lambda: ops.bbb_fused_sum_n(cells, g1, gO, cG, hG, hO, off, 0, True, 0, True, ctx=sc)),
    ("  plain + sum", lambda: ops.bbb_fused(cells, g1, gOe, cG, hG, hOe, off, 0, True, 0, True, ctx=sc)[1].sum(0, keepdim=True)),
]
for name, fn in stages:
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-42s %8.3f ms" % (name, e0.elapsed_time(e1) / 10), flush=True)

#!/usr/bin/env python3
"""Development helper: BASELINE configs[3] (3D smooth-step N=8 C=8 128^3 P=2^19), the four stages with a fresh
StepContext per step; run under rocprofv3 --kernel-trace --stats for per-kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosinesampler_amd import multicell_offset, ops
dev = torch.device("cuda", 0)
N, C, S, P = 8, 8, 128, 1 << 19
torch.manual_seed(0)
c3 = torch.rand(N, C, S, S, S, device=dev)
g3 = torch.rand(N, 1, 1, P, 3, device=dev) * 2 - 1
go3 = torch.randn(N, C, 1, 1, P, device=dev); ho3 = torch.randn_like(go3)
cg3 = torch.randn_like(g3); hg3 = torch.randn_like(g3)
o3 = multicell_offset(N, True, dev)
for step in range(int(os.environ.get("CS_STEPS", "5"))):
    sc = ops.StepContext()
    ops.forward(c3, g3, o3, 0, True, 2, True, ctx=sc)
    ops.backward(go3, c3, g3, o3, 0, True, True, 2, True, ctx=sc)
    if os.environ.get("CS_EXTRA", "1") == "1":      # the same stages without the table gradient: point kernels alone
        ops.backward(go3, c3, g3, o3, 0, True, False, 2, True, ctx=sc)
        ops.backward_backward(None, cg3, c3, g3, go3, o3, 0, True, False, 2, True, ctx=sc, want_grad_input=False)
    ops.backward_backward(None, cg3, c3, g3, go3, o3, 0, True, False, 2, True, ctx=sc)
    ops.bbb_fused(c3, g3, go3, cg3, hg3, ho3, o3, 0, True, 2, True, ctx=sc)
torch.cuda.synchronize()

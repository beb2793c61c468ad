#!/usr/bin/env python3
"""Stage times for a 3D shape: python tools/stage_time_3d.py N C S P [reps]   (default: the reference test_3d.py shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosinesampler_amd import multicell_offset, ops
a = [int(x) for x in sys.argv[1:]]
N, C, S, P = (a + [50, 4, 16, 100000][len(a):])[:4]
reps = a[4] if len(a) > 4 else 8
dev = torch.device("cuda", 0)
torch.manual_seed(0)
c3 = torch.rand(N, C, S, S, S, device=dev)
xyz = torch.rand(P, 3, device=dev) * 2 - 1
g3 = xyz.view(1, 1, 1, P, 3).repeat(N, 1, 1, 1, 1).contiguous()
go3 = torch.randn(N, C, 1, 1, P, device=dev); ho3 = torch.randn_like(go3)
cg3 = torch.randn_like(g3); hg3 = torch.randn_like(g3)
o3 = multicell_offset(N, True, dev)
sc3 = ops.StepContext()
if os.environ.get("CS_FORCE"):
    ops.force_path(int(os.environ["CS_FORCE"]))
st = {
    "3D forward": lambda: ops.forward(c3, g3, o3, 0, True, 2, True, ctx=sc3),
    "3D backward": lambda: ops.backward(go3, c3, g3, o3, 0, True, True, 2, True, ctx=sc3),
    "3D backward (no grad_input)": lambda: ops.backward(go3, c3, g3, o3, 0, True, False, 2, True, ctx=sc3),
    "3D backward_backward": lambda: ops.backward_backward(None, cg3, c3, g3, go3, o3, 0, True, False, 2, True, ctx=sc3),
    "3D bbb_fused": lambda: ops.bbb_fused(c3, g3, go3, cg3, hg3, ho3, o3, 0, True, 2, True, ctx=sc3),
}
print("N=%d C=%d %d^3 P=%d  (S = %.1f M samples)" % (N, C, S, P, N * P / 1e6))
for name, fn in st.items():
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-30s %8.3f ms" % (name, e0.elapsed_time(e1) / reps), flush=True)

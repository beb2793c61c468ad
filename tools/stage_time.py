#!/usr/bin/env python3
"""Development helper: time each stage of the headline config separately (prepared objects prebuilt),
plus the plan build and the channels-last pack.  python tools/stage_time.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosinesampler_amd import multicell_offset, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
N, C, H, P = 16, int(os.environ.get("CS_C", "16")), int(sys.argv[2]) if len(sys.argv) > 2 else 256, 1 << 20
N = int(os.environ.get("CS_N", N)); P = int(os.environ.get("CS_P", P))
print("N=%d C=%d H=W=%d P=%d" % (N, C, H, P))
torch.manual_seed(0)
cells = torch.rand(N, C, H, H, device=dev)
xy = torch.rand(P, 2, device=dev) * 2 - 1
if os.environ.get("CS_SORT"):      # the points in the order ops.sort_points gives them: what a caller that orders its collocation set hands over
    xy, _ = ops.sort_points(xy, (H, H))
grid = xy.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
gOut = torch.randn(N, C, 1, P, device=dev); hO = torch.randn(N, C, 1, P, device=dev)
cG = torch.randn(N, 1, P, 2, device=dev); hG = torch.randn(N, 1, P, 2, device=dev)
off = multicell_offset(N, True, dev)
a = (0, True, 0, True)
if os.environ.get("CS_FORCE"):
    ops.force_path(int(os.environ["CS_FORCE"]))
if os.environ.get("CS_ORDER"):      # 'coherent' / 'random': force the hint (default: measured)
    ops.points_order(os.environ["CS_ORDER"])
if os.environ.get("CS_ABLATE"):     # coherent kernels with parts switched off (cs_debug_coherent_tuning): results are wrong
    ops._lib.load().cs_debug_coherent_tuning(int(os.environ.get("CS_CHUNK", "0")), int(os.environ["CS_ABLATE"]))
sc = ops.StepContext()
stages = {
    "forward": lambda: ops.forward(cells, grid, off, *a, ctx=sc),
    "backward": lambda: ops.backward(gOut, cells, grid, off, 0, True, True, 0, True, ctx=sc),
    "backward(no grad_input)": lambda: ops.backward(gOut, cells, grid, off, 0, True, False, 0, True, ctx=sc),
    "backward_backward": lambda: ops.backward_backward(None, cG, cells, grid, gOut, off, 0, True, False, 0, True, ctx=sc),
    "bbb_fused": lambda: ops.bbb_fused(cells, grid, gOut, cG, hG, hO, off, 0, True, 0, True, ctx=sc),
    "plan+pack (fresh ctx fwd+bwd minus above)": None,
}
for name, fn in stages.items():
    if fn is None:
        continue
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-28s %8.3f ms" % (name, e0.elapsed_time(e1) / reps), flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "3d":
    # BASELINE.json configs[3]: 3D smoothstep N=8 C=8 128^3 P=2^19
    N3, C3, S3, P3 = 8, 8, 128, 1 << 19
    c3 = torch.rand(N3, C3, S3, S3, S3, device=dev)
    g3 = torch.rand(N3, 1, 1, P3, 3, device=dev) * 2 - 1
    go3 = torch.randn(N3, C3, 1, 1, P3, device=dev); ho3 = torch.randn_like(go3)
    cg3 = torch.randn_like(g3); hg3 = torch.randn_like(g3)
    o3 = multicell_offset(N3, True, dev)
    sc3 = ops.StepContext()
    st3 = {
        "3D forward": lambda: ops.forward(c3, g3, o3, 0, True, 2, True, ctx=sc3),
        "3D backward": lambda: ops.backward(go3, c3, g3, o3, 0, True, True, 2, True, ctx=sc3),
        "3D backward_backward": lambda: ops.backward_backward(None, cg3, c3, g3, go3, o3, 0, True, False, 2, True, ctx=sc3),
        "3D bbb_fused": lambda: ops.bbb_fused(c3, g3, go3, cg3, hg3, ho3, o3, 0, True, 2, True, ctx=sc3),
        "3D forward (no ctx: NCDHW gathers)": lambda: ops.forward(c3, g3, o3, 0, True, 2, True),
    }
    for name, fn in st3.items():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("%-28s %8.3f ms" % (name, e0.elapsed_time(e1) / reps), flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    c2 = ops.StepContext()
    lib = ops._lib.load()
    st = torch.cuda.current_stream().cuda_stream
    c2.input_cl(lib, cells, 2, [N, C, H, H], P, st)
    c2.plan(lib, grid, off, 2, [N, C, H, H], P, 0, True, True, st)
e1.record()
torch.cuda.synchronize()
print("%-28s %8.3f ms" % ("pack + plan build", e0.elapsed_time(e1) / reps))

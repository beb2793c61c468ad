import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
for (N, C, H, P) in ((96, 4, 16, 100000), (96, 4, 16, 1000000), (16, 16, 256, 1 << 20)):
    ms = bench.helmholtz_step(N, C, H, P, dev, steps=5)
    print("Helmholtz autograd step N=%d C=%d H=W=%d P=%d: %.3f ms  (%.0f Msamples/s)" % (N, C, H, P, ms, N * P / ms / 1e3), flush=True)

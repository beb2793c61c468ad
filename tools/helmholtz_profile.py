import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
print(bench.helmholtz_step(16, 16, 256, 1 << 20, dev, steps=3))

import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
# HELM=sorted: the collocation points ordered once (ops.sort_points) and handed over as a (1,1,P,2) broadcast grid
kw = dict(sort_points=True, broadcast_grid=True) if os.environ.get("HELM") == "sorted" else {}
print(bench.helmholtz_step(16, 16, 256, 1 << 20, dev, steps=3, **kw))

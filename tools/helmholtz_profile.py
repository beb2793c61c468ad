import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
# HELM=sorted: the collocation points ordered once (ops.sort_points) and handed over as a (1,1,P,2) broadcast grid;
# HELM=summed: ordered points and CosineSampler2dSum (the summing kernels)
# HELM=summed_drawn: CosineSampler2dSum on the points as drawn (the op orders them itself)
kw = {"sorted": dict(sort_points=True, broadcast_grid=True), "summed": dict(sort_points=True, summed_op=True),
      "summed_drawn": dict(summed_op=True)}.get(os.environ.get("HELM"), {})
print(bench.helmholtz_step(16, 16, 256, 1 << 20, dev, steps=3, **kw))

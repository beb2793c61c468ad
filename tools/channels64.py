import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
a = bench.stage_pipeline_ms(dev, 2, 16, 32, 256, 1 << 20, 0, steps=5)[0]
b = bench.stage_pipeline_ms(dev, 2, 16, 64, 256, 1 << 20, 0, steps=5)[0]
print("C=32 %.3f ms  C=64 %.3f ms  ratio to two 32-channel steps %.3f" % (a, b, b / (2 * a)))

import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0)
for srt in (False, True):
    for warm in (False, True):
        print("3D config sorted=%s warm_plan=%s: %.3f ms" % (srt, warm, bench.stage_pipeline_ms(dev, 3, 8, 8, 128, 1 << 19, 2, warm_plan=warm, sort_points=srt)[0]), flush=True)

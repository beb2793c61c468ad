#!/usr/bin/env python3
"""Development helper (GPU box): the four-stage step at the other shapes bench.py reports, for the library COSINESAMPLER_LIB
selects (tools/ab.sh builds the variants) -- bf16 streams, the reference's test shapes, 32 channels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
name = os.path.basename(os.environ.get("COSINESAMPLER_LIB", "default"))
r = []
r.append(("bf16", bench.stage_pipeline_ms(dev, 2, 16, 16, 256, 1 << 20, 0, stream_dtype=torch.bfloat16)[0]))
r.append(("f16", bench.stage_pipeline_ms(dev, 2, 16, 16, 256, 1 << 20, 0, stream_dtype=torch.float16)[0]))
r.append(("fp32", bench.stage_pipeline_ms(dev, 2, 16, 16, 256, 1 << 20, 0)[0]))
r.append(("ref2d N=96 C=4 16^2", bench.stage_pipeline_ms(dev, 2, 96, 4, 16, 100000, 0)[0]))
r.append(("C=32", bench.stage_pipeline_ms(dev, 2, 16, 32, 256, 1 << 20, 0, steps=5)[0]))
r.append(("N=8 C=8 512^2", bench.stage_pipeline_ms(dev, 2, 8, 8, 512, 1 << 20, 2)[0]))
print("%-18s " % name + "  ".join("%s %.3f" % kv for kv in r), flush=True)

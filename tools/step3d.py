#!/usr/bin/env python3
"""Development helper: BASELINE configs[3] (3D smooth-step N=8 C=8 128^3 P=2^19) as bench.py runs it -- fresh StepContext per
step -- with the per-step time; COSINESAMPLER_LIB selects an A/B library (tools/ab.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
ms, _ = bench.stage_pipeline_ms(dev, 3, 8, 8, 128, 1 << 19, 2, steps=10)
ms_s, _ = bench.stage_pipeline_ms(dev, 3, 8, 8, 128, 1 << 19, 2, steps=10, sort_points=True)
ms_u, _ = bench.stage_pipeline_ms(dev, 3, 8, 8, 128, 1 << 19, 2, steps=10, summed=True)
ms_us, _ = bench.stage_pipeline_ms(dev, 3, 8, 8, 128, 1 << 19, 2, steps=10, summed=True, sort_points=True)
print("%-10s config_3d %.3f ms   sorted points %.3f ms   summed op %.3f ms   summed op, sorted points %.3f ms"
      % (os.path.basename(os.environ.get("COSINESAMPLER_LIB", "default")), ms, ms_s, ms_u, ms_us), flush=True)

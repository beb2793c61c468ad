"""CPU: bench.py's arithmetic matches BASELINE.md section 3, and its contract fields exist in the
last recorded GPU run (profiles/round1_bench.json), without needing a GPU."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_match_baseline_md():
    S, C, d, T = 1 << 24, 16, 2, 4 * 16 * 16 * 256 * 256
    ab = bench.algorithmic_bytes(S, C, d, T)
    assert ab == {"forward": 1275068416, "backward": 1476395008, "backward_backward": 2684354560,
                  "bbb_fused": 3758096384}
    assert sum(ab.values()) == 9193914368            # 548 B/sample, BASELINE.md section 3
    S3, C3, T3 = 1 << 22, 8, 4 * 8 * 8 * 128 ** 3     # config 4
    assert sum(bench.algorithmic_bytes(S3, C3, 3, T3).values()) == 5150605312


def test_recorded_bench_line_has_the_contract_fields():
    path = os.path.join(ROOT, "profiles", "round1_bench.json")
    line = json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "Msamples/s" and line["dtype"] == "f32" and line["vs_baseline"] is None
    assert line["scaling"] == "weak" and "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    # value is samples / time
    assert abs(line["value"] - line["config"]["samples_per_step_per_gpu"] * line["n_gpus"] / line["ms_per_step"] / 1e3) < 1.0

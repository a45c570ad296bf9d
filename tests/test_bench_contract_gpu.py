"""bench.py's contract with the driver: ONE JSON line on stdout carrying the named keys, also when the RCCL
path is up (RCCL prints a version banner to descriptor 1 from C) and under torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def test_bench_prints_one_json_line_with_rccl_up():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ, ZERO_AMD_BENCH_WATCHDOG="240")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in REQUIRED:
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["unit"] == "images/sec" and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["dtype"] == "bf16" and out["data"] == "synthetic" and out["vs_baseline"] is None
    assert "workload" in out["config"] and "model" not in out["config"]
    assert abs(out["value"] - out["config"]["global_batch"] / (out["ms_per_step"] * 1e-3)) <= 0.01 * out["value"]
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.2 < rf["frac"] < 1.0
    assert 20.0 < out["value"] < 1000.0

"""bench.py's secondary workloads keep running (the driver only exercises the default C3 line): one short C2 pass
(encoder + adapter fine-tune objective, BASELINE.json configs[1] at a small batch) in a child process, and the JSON
contract of the line it prints."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_c2_line():
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--batch", "8", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0 and line["dtype"] == "bf16"
    assert "workload" in line["config"]

"""bench.py prints one JSON line with the contract's keys (small workload)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


@pytest.mark.gpu
def test_bench_line_small_workload():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--depth", "6", "--size", "320x200", "--cpu-seconds", "0.5"],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    for k in REQUIRED:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["unit"] == "Mray/s" and j["dtype"] == "f32"
    assert j["vs_baseline"] is None and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mray/s" and c["cores"] >= 1 and c["value"] > 0


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode != 0 and "no CPU path" in (out.stderr + out.stdout)

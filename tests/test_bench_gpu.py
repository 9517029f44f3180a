"""bench.py end to end on the GPU box: ONE JSON line with the contract's keys (driver-facing; see the task's bench contract)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*args):
    env = dict(os.environ, SIGNAL_CPU_THREADS="16")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_default_workload_json_contract():
    d = _run("--steps", "3", "--warmup", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("images/sec") and d["unit"] == "triplets/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 64 * 1000.0 / d["ms_per_step"]) / d["value"] < 0.02
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert r["launches"] == 12 * 3 and 0.05 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 1e8
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "triplets/s" and c["value"] > 0 and 1 <= c["cores"] <= 16 and "sample" in c
    assert d["value"] > 50 * c["value"]          # sanity: the HIP path is not a CPU fallback


def test_bench_train_workload_runs():
    d = _run("--workload", "train", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert d["config"]["workload"].startswith("configs[2]") and d["value"] > 0 and "cpu_baseline" not in d
    assert d["roofline"]["launches"] == 12 * 2

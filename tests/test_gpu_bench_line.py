"""The driver-facing contract of `bench.py`, end to end on the GPU: one process, the c2 workload (small enough for the CPU
baseline to take seconds), the single JSON line on stdout with every field the round's measurement rules name."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    env = dict(os.environ, OPENBLAS_NUM_THREADS=os.environ.get("OPENBLAS_NUM_THREADS", "16"))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--workload", "poisson1d"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    line = json.loads(lines[-1])
    assert sum(1 for ln in lines if ln.lstrip().startswith("{")) == 1
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["dtype"] == "f64" and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert "synthetic" in line["data"] and line["config"]["workload"].startswith("poisson1d")
    assert "model" not in line["config"]
    # value = algorithmic GFLOP/s of the whole step
    assert abs(line["value"] * 1e9 * line["ms_per_step"] * 1e-3 / line["config"]["algorithmic_flops_per_step"] - 1.0) < 1e-6
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 78.6
    assert 0.0 < roof["frac"] < 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0
    assert line["parity"]["pass"] is True
    assert len(line["config"]["csrc_sha16"]) == 16

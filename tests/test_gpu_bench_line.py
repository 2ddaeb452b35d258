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
    # round 5: the mode of the timed region is stated, the default mode and the reference's own call sequence are timed beside it
    assert line["config"]["lazy_factorization"] is True and line["config"]["fused_factor_and_predict"] is True
    assert line["modes"]["eager_default"]["ms_per_step"] > 0 and line["modes"]["timed_region"]["lazy_factorization"] is True
    seq = line["reference_sequence"]
    assert "u.mean(x)" in seq["calls"] and "u.std(x)" in seq["calls"]
    assert seq["default_mode_ms"] > 0 and seq["lazy_mode_ms"] > 0
    # (c2's posterior variance is 2.5e-7 of the prior's: k(x,x) - sum v^2 cancels seven digits, and two orders of summation --
    #  the fused pipeline, the blocked substitution of the second pipeline -- differ by the rounding of that difference, a few
    #  1e-9 of the largest variance: the parity bar, not 1e-12; on well-conditioned problems the two agree to 1e-12,
    #  tests/test_gpu_fused.py::test_reference_sequence_mean_then_std)
    assert seq["mean_vs_predict_rel"] <= 1e-12 and seq["var_from_std_vs_predict_rel"] <= 1e-8
    assert line["e2e_with_h2d_ms"] > 0 and line["e2e"]["mean_vs_resident_rel"] <= 1e-12 and line["e2e"]["var_vs_resident_rel"] <= 1e-11
    # round 6: the schedule of the timed region by name, and the two-pipeline step time -- the schedule EVERY N > 1 line runs -- as
    # top-level fields a scaling reader can use
    assert line["mode"] == "fused_factor_and_predict"
    assert line["two_pipeline_ms_per_step"] > 0 and line["two_pipeline_value"] > 0
    assert abs(line["two_pipeline_value"] * line["two_pipeline_ms_per_step"] / (line["value"] * line["ms_per_step"]) - 1.0) < 1e-9
    # the reference's own sequence no longer pays 2 T dependent launches for its weights (trsv.hip): within 25 % of predict at c2
    assert seq["default_mode_ms"] <= 1.25 * line["two_pipeline_ms_per_step"], (seq, line["two_pipeline_ms_per_step"])


def test_plain_call_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the way the driver calls N = 1, the way the reference is
    called: one process, `_conditional.py:253-294`): the process starts its two ranks itself, before any GPU call, and
    the line is a line for TWO ranks -- never a one-GPU line for a two-GPU request.  On the one GPU of the test box both
    ranks open device 0 and RCCL runs over loopback sockets (bring-up aid: code path, not rates)."""
    env = dict(os.environ, LPGP_DEVICE="0", LPGP_BENCH_RCCL_LOOPBACK="1", OPENBLAS_NUM_THREADS="8")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--n-side", "48", "--m-side", "16"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["config"]["rccl_ranks"] == 2 and line["config"]["transport"] == "rccl"
    assert line["config"]["process_grid"] == [2, 1]
    assert len(line["config"]["comm_per_rank_per_step"]) == 2
    assert all(r["bytes_received"] > 0 for r in line["config"]["comm_per_rank_per_step"])
    # round 5: real parity at N > 1 -- the timed workload against the oracle on rank 0's host cores, in the same run
    assert line["parity"]["pass"] is True and "oracle" in line["parity"], line["parity"]
    assert line["config"]["calibration_seconds"] >= 0
    # round 6: every N > 1 line names its schedule, and the field the N = 1 line is to be compared through is the same number here
    assert line["mode"] == "two_pipelines" and abs(line["two_pipeline_ms_per_step"] - line["ms_per_step"]) < 1e-9

"""bench.py end to end on the GPU box: the one-line JSON contract at N = 1 and the multi-rank control flow (two ranks
on the one GPU, gloo rehearsal backend - RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _json_line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "19", "--warmup", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 19 and d["warmup"] == 3 and d["unit"] == "evals/s"
    assert abs(d["value"] - 1e6 * 19 / (d["ms_per_step"] * 19e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel_ms"] > 0
    assert d["check"]["max_abs_err_vs_oracle"] < 1e-10 and d["check"]["gather_ok"]


def test_bench_two_ranks_gloo_rehearsal():
    env = dict(os.environ, ROBCHAR_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "11", "--warmup", "3"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["check"]["gather_ok"] and d["check"]["max_abs_err_vs_oracle"] < 1e-10
    assert abs(d["value"] - 2e6 * 11 / (d["ms_per_step"] * 11e-3)) / d["value"] < 1e-6

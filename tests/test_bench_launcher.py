"""`bench.py --gpus N` without a launcher, on a box WITHOUT a GPU: the parent must start N child ranks (it never touches
the GPU itself), every rank must fail loudly (there is no CPU path), and the launcher must hand back a non-zero exit code
without printing a JSON line.  (The same command on the GPU box: tests/test_gpu_bench.py::test_bench_self_launch_two_ranks.)"""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        return importlib.import_module("code-robchar_amd._lib").load().rc_device_count() > 0
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="control-flow test for a box without a GPU")
def test_self_launch_fails_loudly_without_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["ROBCHAR_BENCH_RANK_GRACE_S"] = "5"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert r.stdout.strip() == ""                                   # no line: nothing was measured
    assert r.stderr.count("bench.py needs a GPU") == 3              # three ranks were started, each said why it stops
    assert "rank exit codes [1, 1, 1]" in r.stderr


def test_launcher_does_not_touch_torch_in_the_parent():
    """The parent of a self-launched run must not initialise the GPU (a later spawn from a GPU-initialised process is what
    takes machines down on this pool): `launch_ranks` is reached before `Env` and before any torch import."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args.gpus)") < main.index("Env(args)")
    assert main.index("launch_ranks(args.gpus)") < main.index("cpu_baseline(cfg")
    head = src[:src.index("def main():")]
    launcher = head[head.index("def launch_ranks("):head.index("def cold_kernel_ms(")]
    code = launcher[launcher.index('"""', launcher.index('"""') + 3) + 3:]          # (the docstring mentions torch.distributed.run)
    assert "import torch" not in code and "torch." not in code
    top_level_imports = [l for l in src.splitlines() if l.startswith("import ") or l.startswith("from ")]
    assert not any("torch" in l for l in top_level_imports)

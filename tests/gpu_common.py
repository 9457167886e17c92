"""Helpers shared by the GPU test modules (tests/test_gpu_*.py): synthetic controllers, matrix injection through the draws,
the dense layout of directional samples, generator-state comparison."""
import numpy as np

from oracle import robchar_oracle as orc


def rand_ctrl(rng, C, N):
    x = np.empty((C, N + 1))
    x[:, :N] = rng.uniform(-10, 10, (C, N))
    x[:, N] = rng.uniform(2, 30, C)
    return x


def _h0(case):
    return orc.xxz_delta(case["N"]) if case["mode"] == "xxz" else None


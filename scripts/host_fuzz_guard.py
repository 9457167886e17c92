#!/usr/bin/env python3
"""Host-side calibration of the a-posteriori sum-rule guard (tridiag_core.h: kSumRuleGuard): the per-sample arithmetic of the
chain kernels compiled for the CPU (tests/host/host_core.cpp, one sample = one "wave"), the adversarial configuration
generator of scripts/fuzz_parity.py, guard on / off.  Prints the worst |dF| against the oracle per weight mode and how
many samples each build sent to the eigenvector route, plus the flag rate on the benchmark workloads (false positives
cost time there).  Development aid; needs no GPU.   usage: host_fuzz_guard.py [nseeds] [ncfg]"""
import ctypes, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import robchar_oracle as orc
P = ctypes.POINTER(ctypes.c_double)


def build(flags):
    out = os.path.join(tempfile.mkdtemp(prefix="rc_hostfuzz_"), "lib.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC"] + flags + ["-o", out,
                    os.path.join(ROOT, "tests", "host", "host_core.cpp")], check=True)
    lib = ctypes.CDLL(out)
    lib.rc_host_general_calls.restype = ctypes.c_longlong

    def fid(ctrl, draws, N, a, b, h0d=None, variant=0):
        lib.rc_host_set_variant(variant)
        C, K = draws.shape[:2]
        ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        h0d = np.zeros(N) if h0d is None else np.ascontiguousarray(h0d, dtype=np.float64)
        h0o = np.ones(max(N - 1, 1))
        res = np.empty((C, K))
        assert lib.rc_host_chain_fidelity(N, ctrl.ctypes.data_as(P), h0d.ctypes.data_as(P), h0o.ctypes.data_as(P),
                                          draws.ctypes.data_as(P), ctypes.c_longlong(C), ctypes.c_longlong(K), a, b,
                                          res.ctypes.data_as(P)) == 0
        return res
    fid.calls = lib.rc_host_general_calls
    return fid


def configs(seed, ncfg):
    rng = np.random.default_rng(seed)
    for it in range(ncfg):
        N = int(rng.integers(2, 17))
        C, K = int(rng.integers(1, 6)), int(rng.integers(1, 700))
        amp = float(rng.choice([1.0, 10.0, 100.0]))
        sig = float(rng.choice([0.0, 1e-3, 0.05, 0.2, 0.5]))
        ctrl = np.empty((C, N + 1))
        ctrl[:, :N] = rng.uniform(-amp, amp, (C, N))
        ctrl[:, N] = rng.uniform(0.0, float(rng.choice([1.0, 30.0, 100.0])), C) * rng.choice([-1, 1], C)
        draws = sig * rng.standard_normal((C, K, N, 3))
        mode = rng.random()
        if mode < 0.15 and N >= 3:
            i, j = sorted(rng.choice(N, 2, replace=False))
            ctrl[:, j] = ctrl[:, i] + 10.0 ** rng.uniform(-9, -2) * rng.choice([-1, 1], C)
        elif mode < 0.22 and N >= 4:
            cut = N // 2
            ctrl[:, N - cut:N] = ctrl[:, :cut][:, ::-1]
            draws[:, ::3, :, 0] = 0.0
            draws[:, ::3, cut, 1], draws[:, ::3, cut, 2] = -1.0, 0.0
            for q in range(1, cut):
                draws[:, ::3, N - q, 1:] = draws[:, ::3, q, 1:]
        elif mode < 0.30 and N >= 4:
            cut = N // 2
            ctrl[:, N - cut:N] = ctrl[:, :cut][:, ::-1]
            draws[:, ::2, :, 0] *= 1e-3
            draws[:, ::2, cut, 1], draws[:, ::2, cut, 2] = -1.0 + 10.0 ** rng.uniform(-9, -2), 0.0
            for q in range(1, cut):
                draws[:, ::2, N - q, 1:] = draws[:, ::2, q, 1:]
        elif mode < 0.36:
            ctrl[:, :N] = rng.uniform(-1e-6, 1e-6, (C, N)) + rng.uniform(-amp, amp)
            draws *= 10.0 ** rng.uniform(-8, -2)
        h0 = orc.xxz_delta(N) if rng.random() < 0.3 else None
        a, b = int(rng.integers(0, N)), int(rng.integers(0, N))
        if rng.random() < 0.4:
            a, b = 0, N - 1
        yield dict(seed=seed, it=it, N=N, amp=amp, sig=sig, a=a, b=b, xxz=h0 is not None), ctrl, draws, h0


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ncfg = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    extra = os.environ.get("HOSTFUZZ_FLAGS", "").split()
    libs = {"guard": build(["-DRC_SUM_RULE_GUARD=1"] + extra), "guard m0": build(["-DRC_SUM_RULE_GUARD=1", "-DRC_SUM_RULE_MOMENTS=1"] + extra), "no guard": build(["-DRC_SUM_RULE_GUARD=0"] + extra)}
    worst = {k: {0: (0, None), 2: (0, None)} for k in libs}
    nsamp = 0
    t0 = time.time()
    for seed in range(5000, 5000 + nseeds):
        for meta, ctrl, draws, h0 in configs(seed, ncfg):
            want = orc.fidelity_eigh(ctrl, draws, meta["N"], meta["a"], meta["b"], h0_diag=h0)
            nsamp += want.size
            for name, f in libs.items():
                for variant in (0, 2):             # 0: ends when applicable, else adjugate; 2: general adjugate always
                    e = float(np.abs(f(ctrl, draws, meta["N"], meta["a"], meta["b"], h0, variant) - want).max())
                    if e > worst[name][variant][0]:
                        worst[name][variant] = (e, meta)
    print(f"{nseeds * ncfg} adversarial configurations, {nsamp} samples x 2 weight-mode variants, {time.time() - t0:.0f} s")
    for name, f in libs.items():
        print(f"{name:9s}: samples on the eigenvector route {f.calls()} of {2 * nsamp}")
        for variant, label in ((0, "auto (ends / adjugate)"), (2, "general adjugate")):
            print(f"           {label:24s} worst |dF| = {worst[name][variant][0]:.2e} at {worst[name][variant][1]}")
    # false positives on the benchmark workloads (uniform random controllers, sigma = 0.05): every flagged sample is wasted time
    for (N, a, b, xxz, label) in ((7, 0, 6, False, "config 3"), (7, 0, 3, False, "config 4"), (10, 0, 9, True, "config 5"),
                                  (5, 0, 4, False, "config 2"), (5, 0, 2, False, "config 1"), (13, 3, 9, False, "N = 13 adjugate")):
        rng = np.random.default_rng(20220714 + N)
        C, K = 100, 400
        ctrl = np.empty((C, N + 1))
        ctrl[:, :N] = rng.uniform(-10, 10, (C, N))
        ctrl[:, N] = rng.uniform(2, 30, C)
        draws = 0.05 * rng.standard_normal((C, K, N, 3))
        h0 = orc.xxz_delta(N) if xxz else None
        want = orc.fidelity_eigh(ctrl, draws, N, a, b, h0_diag=h0)
        row = []
        for name, f in libs.items():
            c0 = f.calls()
            e = float(np.abs(f(ctrl, draws, N, a, b, h0, 0) - want).max())
            row.append(f"{name}: {f.calls() - c0} flagged, max |dF| {e:.1e}")
        print(f"{label:16s} {C * K} samples   " + "   ".join(row))


if __name__ == "__main__":
    main()

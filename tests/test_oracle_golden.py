"""Pins the CPU oracle (oracle/robchar_oracle.py) against the golden vectors (CPU only).

Sources of the vectors: tests/golden/make_golden.py (outputs of the unmodified reference run in the build
container) and reference-authored data (shipped sigma_sim = 0 caches, L-BFGS best_fid records, the `wd`
test vector of wd_sortof_fast_implementation.py:184-185, the Envtest known answers).
"""
import json

import numpy as np
import pytest

from conftest import load_json
from oracle import robchar_oracle as orc

TOL = 1e-12     # oracle vs reference outputs (measured: <= 2e-13)


def _h0(case):
    return orc.xxz_delta(case["N"]) if case["mode"] == "xxz" else None


@pytest.mark.parametrize("route", ["eigh", "expm"])
def test_kernel_cases(kernel_cases, route):
    fn = orc.fidelity_eigh if route == "eigh" else orc.fidelity_expm_loop
    worst = 0.0
    for case in kernel_cases:
        if route == "expm" and case["N"] not in (4, 5, 7):
            continue
        for s in range(case["draws"].shape[0]):
            got = fn(case["ctrl"], case["draws"][s], case["N"], case["inspin"], case["outspin"],
                     h0_diag=_h0(case), ring=case["mode"] == "ring")
            worst = max(worst, np.abs(got - case["fid"][s]).max())
    assert worst < TOL, worst


def test_sigma0_matches_noiseless(kernel_cases):
    for case in kernel_cases:
        got = orc.fidelity_eigh(case["ctrl"], None, case["N"], case["inspin"], case["outspin"],
                                h0_diag=_h0(case), ring=case["mode"] == "ring")
        assert np.abs(got[:, :1] - case["fid"][0]).max() < TOL


def test_shipped_sigma0_rows(shipped_sigma0):
    """Reference-authored: shipped .le controllers -> sigma_sim=0 row of the shipped .mc caches."""
    n = 0
    for sl in shipped_sigma0:
        got = orc.fidelity_eigh(sl["ctrl"], None, sl["N"], sl["inspin"], sl["outspin"])[:, 0]
        ok = ~np.isnan(sl["fid"])
        assert np.abs(got[ok] - sl["fid"][ok]).max() < 1e-12, sl["name"]
        n += ok.sum()
    assert n > 2000


def test_lbfgs_n7_best_fid(lbfgs_n7):
    """Reference-authored N = 7 known answers: the optimiser's own noiseless fidelity record."""
    for tag, out in (("0-3", 3), ("0-6", 6)):
        got = orc.fidelity_eigh(lbfgs_n7[f"ctrl_{tag}"], None, 7, 0, out)[:, 0]
        assert np.abs(got - lbfgs_n7[f"best_fid_{tag}"]).max() < 1e-11


def test_envtest_known_answers():
    for c in load_json("envtest.json"):
        x = np.array([c["controller"]])
        f = orc.fidelity_eigh(x, None, c["Nspin"], c["inspin"], c["outspin"])[0, 0]
        assert abs(f - c["fid_reference_noise_model"]) < TOL
        if c["envtest_kind"] == "almost":          # assertAlmostEqual(places=2)
            assert round(abs(f - c["envtest_value"]), 2) == 0
        else:
            assert f < c["envtest_value"]


def test_metrics_values():
    g = load_json("metrics.json")
    for k, vec in g["vectors"].items():
        v = g["values"][k]
        a = np.array(vec, dtype=np.float64)
        assert abs(orc.wd_from_ideal(a.copy()) - v["wd_from_ideal"]) < 1e-15
        assert abs(orc.wd_from_ideal_zero(a.copy()) - v["wd_from_ideal_zero"]) < 1e-15
        for p in (0, 1, 2, 3):
            assert abs(orc.rim_p(a, p) - v[f"RIM_{p}"]) < 1e-15
    # published constants quoted in SURVEY.md 8(c)
    assert abs(g["values"]["wd_test_vector"]["wd_from_ideal"] - 0.507069833) < 1e-12
    assert abs(g["values"]["wd_test_vector"]["RIM_2"] - 0.5748732125076583) < 1e-15
    for key, val in g["dkw"].items():
        al, n = key.split("_")
        assert abs(orc.compute_dkw_error(float(al), int(n)) - val) < 1e-16
    assert abs(orc.compute_dkw_error(0.05, 100) - 0.13581015157406195) < 1e-16
    lo, up = orc.dkw_ecdf_bounds(g["vectors"]["wd_test_vector"], 0.95)
    assert np.allclose(lo, g["dkw_bounds_X_0.95"]["lower"], atol=1e-16, rtol=0)
    assert np.allclose(up, g["dkw_bounds_X_0.95"]["upper"], atol=1e-16, rtol=0)
    with pytest.raises(AssertionError):
        orc.wd_from_ideal([0.5, 1.2e9])


def test_slab_metrics_with_nan_row():
    g = load_json("metrics.json")
    slab = np.array(g["slab"], dtype=np.float64)
    rows = orc.metric_rows(slab)
    for name, want in g["slab_metrics"].items():
        assert np.allclose(rows[name], np.array(want), atol=1e-15, rtol=0, equal_nan=True), name


def test_mcsim_seeded_run():
    """Draw order, burn-one-draw-per-level, NaN padding, lbfgs keying, .mcm layout."""
    g = load_json("mcsim_run.json")
    n, a, b = g["Nspin"], g["inspin"], g["outspin"]
    numc, K, noises = g["numcontrollers"], g["bootreps"], np.array(g["noises"])
    for run in g["runs"]:
        tn = run["training_noise"]
        mc = {k: json.loads(v) for k, v in run["files"].items()}
        mcfile = [k for k in mc if k.endswith(".mc")][0]
        np.random.seed(run["seed"])
        want = mc[mcfile]
        if tn is None:
            algos = ["lbfgs"]
        else:
            algos = run["algos"]
        assert "snob" not in run["algos"]
        got = {}
        for algo in algos:
            key = str(n) if algo == "lbfgs" else str(tn)
            got[algo] = orc.mc_fid_tensor(g["le"][algo][key]["controller"], numc, noises, K, n, a, b)
        if tn is not None:
            # get_metrics_dict ran get_fid_dists first; the later get_fid_dists call was a cache hit
            pass
        assert abs(np.random.normal() - run["rng_after"]) < 1e-15, "RNG stream position differs"
        assert set(want.keys()) == set(got.keys())
        for algo in got:
            w = np.array(want[algo], dtype=np.float64)
            assert w.shape == (len(noises), numc, K)
            assert np.array_equal(np.isnan(w), np.isnan(got[algo]))
            assert np.nanmax(np.abs(w - got[algo])) < TOL
        mcm = [k for k in mc if k.endswith(".mcm")]
        if mcm:
            wantm = mc[mcm[0]]
            alpha = 1 - 0.95
            for algo in got:
                gm = orc.metrics_dict_for_tensor(got[algo], alpha)
                assert set(gm.keys()) == set(wantm[algo].keys())
                for name in gm:
                    assert np.allclose(np.array(gm[name]), np.array(wantm[algo][name], dtype=np.float64),
                                       atol=1e-12, rtol=0, equal_nan=True), (algo, name)


def test_get_rims_seeded():
    g = load_json("get_rims.json")
    np.random.seed(g["seed"])
    for cont, want in zip(g["controllers"], g["rims"]):
        got = orc.rims_for_controller(cont, g["noises"], g["bootreps"], g["Nspin"], g["inspin"], g["outspin"])
        assert np.abs(got - np.array(want)).max() < TOL
    assert abs(np.random.normal() - g["rng_after"]) < 1e-15


def test_directional_golden():
    """The reference's `directional_perturbation` (seeded): direction list, layout mapping and the non-Hermitian
    diagonal directions, through the oracle's expm path."""
    g = load_json("directional.json")
    for case in g["cases"]:
        n, C, K = case["Nspin"], case["C"], case["K"]
        assert [list(d) for d in orc.directional_directions(n)] == case["directions"]
        draws = np.zeros((C, K, n, 3))
        imag = np.zeros((C, K, n))
        for s, (idx, (a, b)) in enumerate(zip(case["index"], case["ab"])):
            draws[s // K, s % K], imag[s // K, s % K] = orc.directional_to_layout(n, idx, a, b)
        got = orc.fidelity_expm_loop(np.array(case["controllers"]), draws, n, case["inspin"], case["outspin"],
                                     diag_imag=imag)
        assert np.abs(got - np.array(case["fid"])).max() < 1e-12
        assert imag.any()                          # the non-Hermitian branch is exercised


def test_get_arims_seeded():
    """`NStochOpt.get_arims` of the unmodified reference (seeded): ARIM array, kept checkpoint keys, the popped short
    checkpoint and the RNG position afterwards."""
    g = load_json("get_arims.json")
    cdict = json.loads(json.dumps(g["cdict"]))
    for run in g["runs"]:
        np.random.seed(run["seed"])
        arims, keys = orc.arims_for_checkpoints(cdict[run["algo"]]["0.01"], g["numcontrollers"], g["noises"],
                                                g["bootreps"], g["Nspin"], g["inspin"], g["outspin"])
        assert keys == run["keys"]
        assert np.abs(arims - np.array(run["arims"])).max() < TOL
        assert abs(np.random.normal() - run["rng_after"]) < 1e-15
    assert cdict == g["cdict_after"]


def test_fidelity_ss_av_golden():
    """The optimiser-side objective of `qnewton.LBFGS`: fixed Hamiltonian sets from `np.random.seed(4)` (two real draws
    per site), mean fidelity over the first `reps` train Hamiltonians / the whole 10 000-Hamiltonian test set."""
    g = load_json("fidelity_ss_av.json")
    for c in g["cases"]:
        n = c["Nspin"]
        h0 = orc.xxz_delta(n) if c["heisenberg_int"] else None
        assert np.allclose(c["HH_diag"], h0 if h0 is not None else np.zeros(n), atol=0)
        assert c["max_imag"] == 0.0
        train, test = orc.rand_hset_draws(n, c["sigma"], c["train_size"], c["test_size"])
        # the fixture holds H - HH, i.e. (1 + g) - 1 on the couplings: equal up to one rounding of 1 + g
        close = lambda u, v: np.abs(u - np.array(v)).max() < 5e-16
        assert close(train[:, :, 0], c["train_diag"]) and close(train[:, 1:, 1], c["train_sub"])
        assert close(test[:4, :, 0], c["test_diag_head"]) and close(test[:4, 1:, 1], c["test_sub_head"])
        for i, x in enumerate(c["controllers"]):
            a = (n, c["inspin"], c["outspin"])
            assert abs(orc.fidelity_ss_av(x, train, *a, reps=c["reps"], h0_diag=h0) - c["av_train"][i]) < TOL
            assert abs(orc.fidelity_ss_av(x, train, *a, reps=c["train_size"], h0_diag=h0) - c["av_train_all"][i]) < TOL
            assert abs(orc.fidelity_ss_av(x, test, *a, h0_diag=h0) - c["av_test"][i]) < TOL
            assert abs(orc.fidelity_eigh(np.array([x]), None, *a, h0_diag=h0)[0, 0] - c["noiseless"][i]) < TOL


@pytest.mark.parametrize("cid", [2, 3, 4, 5])
def test_highfid_workloads_pin_the_oracle_where_fidelities_are_large(highfid, cid):
    """Round 5: the oracle against the REFERENCE's own outputs on the delocalised controller sets of the full-size GPU parity
    tests (tests/golden/highfid.npz: shipped N = 5 / N = 7 L-BFGS controllers, constructed N = 10 XXZ ones; four injected
    draws at sigma 0.05 per controller + the noiseless value) - fidelities of O(1), so the bound is also a RELATIVE one -
    and, for config 2, against the shipped cache's sigma_sim = 0 row."""
    from conftest import highfid_workload
    N, a, b, ctrl, h0 = highfid_workload(cid)
    tag = f"c{cid}"
    want = highfid[tag + "_fid"]
    got = orc.fidelity_eigh(ctrl, highfid[tag + "_draws"], N, a, b, h0_diag=h0)
    assert want.mean() > 0.4 and (want > 1e-3).mean() > 0.95            # the teeth
    assert np.abs(got - want).max() < TOL
    big = want > 1e-3
    assert (np.abs(got - want)[big] / want[big]).max() < 1e-11
    got0 = orc.fidelity_eigh(ctrl, None, N, a, b, h0_diag=h0)[:, 0]
    assert np.abs(got0 - highfid[tag + "_fid_noiseless"]).max() < TOL
    if cid == 2:
        assert np.abs(got0 - highfid["c2_shipped_sigma0_fid"]).max() < TOL
    if cid in (3, 4):                                                      # the optimiser's own record of the noiseless fidelity
        from conftest import load_npz
        assert np.abs(got0 - load_npz("lbfgs_n7.npz")["best_fid_0-6" if cid == 3 else "best_fid_0-3"]).max() < 1e-9

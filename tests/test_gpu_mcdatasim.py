"""The product surface: `MCDataSim` end to end on the GPU against the seeded run of the unmodified reference (cache
formats, legacy host / device draws, Philox modes, level batching, ARIM, the optimiser-side objective, the scalar
API), two ranks (gloo) and one rank (RCCL), the product-level fuzz."""
import ctypes
import importlib
import json
import os
import pickle

import numpy as np
import pytest

from conftest import highfid_workload, load_json
from gpu_common import rand_ctrl
from oracle import philox_host
from oracle import robchar_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10


# ----------------------------------------------------------------------------------------------------------------
# device-resident MCDataSim
# ----------------------------------------------------------------------------------------------------------------
def _write_le(g, name="golden"):
    os.makedirs(f"experiments/{name}", exist_ok=True)
    base = f"experiments/{name}/ppo_spin_{g['Nspin']}_{g['inspin']}-{g['outspin']}_c_{g['numcontrollers']}"
    json.dump(g["le"], open(base + ".le", "w"))


def test_mcdatasim_seeded_run_on_gpu(workdir):
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    g = load_json("mcsim_run.json")
    os.makedirs("experiments/golden")
    base = f"experiments/golden/ppo_spin_{g['Nspin']}_{g['inspin']}-{g['outspin']}_c_{g['numcontrollers']}"
    json.dump(g["le"], open(base + ".le", "w"))
    for run in g["runs"]:
        tn = run["training_noise"]
        np.random.seed(run["seed"])
        sim = mcmod.MCDataSim(experiment_name="golden", Nspin=g["Nspin"], inspin=g["inspin"],
                              outspin=g["outspin"], noises=np.array(g["noises"]), bootreps=g["bootreps"],
                              training_noise=tn, numcontrollers=g["numcontrollers"], filemarker=".le",
                              verbose=False)
        if tn is None:
            sim.get_fid_dists(algoname="lbfgs")
        else:
            sim.get_metrics_dict()
        assert abs(np.random.normal() - run["rng_after"]) < 1e-15
        for fname, text in run["files"].items():
            want = json.loads(text)
            got = json.load(open(os.path.join("experiments/golden", fname)))
            assert list(got.keys()) == list(want.keys())
            for algo in want:
                if fname.endswith(".mcm"):
                    assert list(got[algo].keys()) == list(want[algo].keys())
                    for name in want[algo]:
                        assert np.allclose(np.array(got[algo][name], dtype=float),
                                           np.array(want[algo][name], dtype=float), atol=TOL, rtol=0,
                                           equal_nan=True), (algo, name)
                else:
                    w, h = np.array(want[algo], dtype=float), np.array(got[algo], dtype=float)
                    assert np.array_equal(np.isnan(w), np.isnan(h))
                    assert np.nanmax(np.abs(w - h)) < TOL
        for f in os.listdir("experiments/golden"):
            if ".mc" in f:
                os.remove(os.path.join("experiments/golden", f))


def test_get_rims_and_single_sample_api(workdir):
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    g = load_json("get_rims.json")
    sim = mcmod.MCDataSim(experiment_name="r", Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"],
                          noises=np.array(g["noises"]), bootreps=g["bootreps"], numcontrollers=1, verbose=False)
    np.random.seed(g["seed"])
    for cont, want in zip(g["controllers"], g["rims"]):
        assert np.abs(sim.get_rims(cont) - np.array(want)).max() < TOL
    assert abs(np.random.normal() - g["rng_after"]) < 1e-15
    # reference-style scalar loop through the same object (gen_fig_8_arim_fcall_scaling.py:121-132)
    # (the scalar API looks ahead - one launch per block of samples - and must leave numpy's stream exactly where the
    # reference's loop leaves it: all three controllers in sequence, then the recorded next draw)
    np.random.seed(g["seed"])
    for cont, want in zip(g["controllers"], g["rims"]):
        for i, nl in enumerate(g["noises"]):
            sim.noise_model.rng(scale=nl)
            f = sum(sim.noise_model.evaluate_noisy_fidelity(cont, ham_noisy=True) for _ in range(g["bootreps"]))
            assert abs((1 - f / g["bootreps"]) - want[i]) < TOL
    assert abs(np.random.normal() - g["rng_after"]) < 1e-15
    # noiseless call and the XXZ route through the public HH attribute
    for c in load_json("envtest.json"):
        noise = importlib.import_module("code-robchar_amd.noise")
        nm = noise.structured_perturbation(Nspin=c["Nspin"], inspin=c["inspin"], outspin=c["outspin"])
        assert abs(nm.evaluate_noisy_fidelity(np.array(c["controller"])) - c["fid_reference_noise_model"]) < TOL
        nm.HH = nm.HH + np.diag(orc.xxz_delta(c["Nspin"]))
        want = orc.fidelity_eigh(np.array([c["controller"]]), None, c["Nspin"], c["inspin"], c["outspin"],
                                 h0_diag=orc.xxz_delta(c["Nspin"]))[0, 0]
        assert abs(nm.evaluate_noisy_fidelity(np.array(c["controller"])) - want) < TOL


def test_mcdatasim_philox_mode(workdir):
    """Device-generated draws (non-reference RNG mode): the driver's tensor equals the oracle evaluated on the
    host-regenerated Philox stream (oracle/philox_host.py), level blocks laid out consecutively; NaN padding kept."""
    from oracle import philox_host
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    g = load_json("mcsim_run.json")
    os.makedirs("experiments/golden")
    base = f"experiments/golden/ppo_spin_{g['Nspin']}_{g['inspin']}-{g['outspin']}_c_{g['numcontrollers']}"
    json.dump(g["le"], open(base + ".le", "w"))
    N, K, numc = g["Nspin"], 37, g["numcontrollers"]
    noises = np.array([0.0, 0.03, 0.1])
    sim = mcmod.MCDataSim(experiment_name="golden", Nspin=N, inspin=g["inspin"], outspin=g["outspin"], noises=noises,
                          bootreps=K, training_noise=0.05, numcontrollers=numc, filemarker=".le", verbose=False,
                          rng_mode="philox", seed=4242)
    fids = sim.get_fid_dists(algoname="ppo")
    got = np.array(fids["ppo"], dtype=float)
    rows = np.array(g["le"]["ppo"]["0.05"]["controller"])[:numc]
    nvalid = len(rows)
    assert got.shape == (3, numc, K) and np.isnan(got[:, nvalid:]).all()
    off = 0
    for j, sg in enumerate(noises):
        n = nvalid * K * N * 3
        draws = philox_host.philox_normal(4242, off, n, sg).reshape(nvalid, K, N, 3)
        off += n
        want = orc.fidelity_eigh(rows, draws, N, g["inspin"], g["outspin"])
        assert np.abs(got[j, :nvalid] - want).max() < TOL


def _two_rank_worker(rank, world, port, tmp, root):
    import sys
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(tmp)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)                       # both ranks share the one GPU of the test box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    g = json.load(open(os.path.join(root, "tests", "golden", "mcsim_run.json")))
    run = g["runs"][0]
    np.random.seed(run["seed"])
    sim = mcmod.MCDataSim(experiment_name="golden", Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"],
                          noises=np.array(g["noises"]), bootreps=g["bootreps"], training_noise=run["training_noise"],
                          numcontrollers=g["numcontrollers"], filemarker=".le", verbose=False)
    dist.barrier()
    metrics = sim.get_metrics_dict()                 # cold: MC on every rank's shard, metric rows all-gathered
    fids = sim.get_fid_dists()                       # rank 0: cache hit; the others: the gathered tensors again
    # rank 0 alone advanced the reference's stream; its final state was broadcast to every rank
    assert abs(np.random.normal() - run["rng_after"]) < 1e-15, "RNG stream position differs on rank %d" % rank
    mcfile = [k for k in run["files"] if k.endswith(".mc")][0]
    want = json.loads(run["files"][mcfile])
    for algo in want:
        w, h = np.array(want[algo], dtype=float), np.array(fids[algo], dtype=float)
        assert np.array_equal(np.isnan(w), np.isnan(h))
        assert np.nanmax(np.abs(w - h)) < 1e-10
    wantm = json.loads(run["files"][mcfile + "m"])
    for algo in wantm:
        for name in wantm[algo]:
            assert np.allclose(np.array(metrics[algo][name], dtype=float), np.array(wantm[algo][name], dtype=float),
                               atol=1e-10, rtol=0, equal_nan=True), (algo, name)
    dist.barrier()
    dist.destroy_process_group()


def test_mcdatasim_two_ranks_on_gpu(tmp_path):
    """The sharded driver on the HIP path: two ranks (gloo rendezvous, both on GPU 0) split the controllers of every
    sigma level, all-gather, and reproduce the reference's seeded run; only rank 0 writes the cache."""
    import socket
    import torch.multiprocessing as mp
    g = load_json("mcsim_run.json")
    os.makedirs(tmp_path / "experiments" / "golden")
    base = tmp_path / "experiments" / "golden" / f"ppo_spin_{g['Nspin']}_{g['inspin']}-{g['outspin']}_c_{g['numcontrollers']}.le"
    json.dump(g["le"], open(base, "w"))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), root), nprocs=2, join=True)
    files = [f for f in os.listdir(tmp_path / "experiments" / "golden") if f.endswith(".mc")]
    assert len(files) == 1


def _one_rank_nccl_worker(rank, port, tmp, root):
    import sys
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ROBCHAR_FORCE_DIST="1")
    os.chdir(tmp)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    g = json.load(open(os.path.join(root, "tests", "golden", "mcsim_run.json")))
    run = g["runs"][0]
    for mode, kw in (("legacy-device", {}), ("legacy-host", {"legacy_draws": "host"}),
                     ("philox", {"rng_mode": "philox", "seed": 3})):
        np.random.seed(run["seed"])
        sim = mcmod.MCDataSim(experiment_name="golden", Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"],
                              noises=np.array(g["noises"]), bootreps=g["bootreps"], training_noise=run["training_noise"],
                              numcontrollers=g["numcontrollers"], filemarker=".le", verbose=False, **kw)
        assert sim._dist() is not None and sim._dist().get_backend() == "nccl"
        metrics = sim.get_metrics_dict()
        fids = sim.get_fid_dists()
        if mode != "philox":
            assert abs(np.random.normal() - run["rng_after"]) < 1e-15
            mcfile = [k for k in run["files"] if k.endswith(".mc")][0]
            want, wantm = json.loads(run["files"][mcfile]), json.loads(run["files"][mcfile + "m"])
            for algo in want:
                assert np.allclose(np.array(fids[algo], dtype=float), np.array(want[algo], dtype=float), atol=1e-10,
                                   rtol=0, equal_nan=True)
                for name in wantm[algo]:
                    assert np.allclose(np.array(metrics[algo][name], dtype=float), np.array(wantm[algo][name], dtype=float),
                                       atol=1e-10, rtol=0, equal_nan=True), (mode, algo, name)
        for f in os.listdir("experiments/golden"):
            if ".mc" in f:
                os.remove(os.path.join("experiments/golden", f))
    dist.destroy_process_group()


def test_mcdatasim_sharded_path_on_rccl_one_rank(tmp_path):
    """The sharded `MCDataSim` code path on the RCCL backend itself (one rank: RCCL refuses two ranks on one device):
    device scatter of the legacy draws, all-gather of metric rows and fidelity slabs, broadcast of the generator state."""
    import socket
    import torch.multiprocessing as mp
    g = load_json("mcsim_run.json")
    os.makedirs(tmp_path / "experiments" / "golden")
    base = tmp_path / "experiments" / "golden" / f"ppo_spin_{g['Nspin']}_{g['inspin']}-{g['outspin']}_c_{g['numcontrollers']}.le"
    json.dump(g["le"], open(base, "w"))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mp.spawn(_one_rank_nccl_worker, args=(port, str(tmp_path), root), nprocs=1, join=True)


def test_mcdatasim_cache_formats_and_lazy_fids(workdir):
    """`cache_format`: json (reference-readable), npy sidecars + index, none (metrics only, nothing but the metric rows
    leaves the GPU) - all three give the reference's seeded metrics; the fidelity handle is lazy."""
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    g = load_json("mcsim_run.json")
    run = g["runs"][0]
    want_m = json.loads([t for f, t in run["files"].items() if f.endswith(".mcm")][0])
    want_f = json.loads([t for f, t in run["files"].items() if f.endswith(".mc")][0])
    for fmt in ("json", "npy", "none"):
        _write_le(g, fmt)
        np.random.seed(run["seed"])
        sim = mcmod.MCDataSim(experiment_name=fmt, Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"],
                              noises=np.array(g["noises"]), bootreps=g["bootreps"], training_noise=run["training_noise"],
                              numcontrollers=g["numcontrollers"], filemarker=".le", verbose=False, cache_format=fmt)
        met = sim.get_metrics_dict()
        assert abs(np.random.normal() - run["rng_after"]) < 1e-15
        for algo in want_m:
            assert list(met[algo].keys()) == list(want_m[algo].keys())
            for name in want_m[algo]:
                assert np.allclose(np.array(met[algo][name], dtype=float), np.array(want_m[algo][name], dtype=float),
                                   atol=TOL, rtol=0, equal_nan=True), (fmt, algo, name)
        files = sorted(os.listdir(f"experiments/{fmt}"))
        mc = [f for f in files if f.endswith(".mc")]
        assert len([f for f in files if f.endswith(".mcm")]) == 1
        if fmt == "none":
            assert not mc and not [f for f in files if f.endswith(".npy")]
            handle = sim._metric_rows["ppo"][0]
            assert handle._host is None                                  # nothing was copied back
            assert np.allclose(np.array(handle), np.array(want_f["ppo"], dtype=float), atol=TOL, equal_nan=True)
            continue
        assert len(mc) == 1
        loaded = sim.loadsimdata(os.path.join(f"experiments/{fmt}", mc[0]))
        raw = json.load(open(os.path.join(f"experiments/{fmt}", mc[0])))
        assert (fmt == "npy") == ("__robchar_npy__" in raw)
        for algo in want_f:
            assert np.allclose(np.array(loaded[algo], dtype=float), np.array(want_f[algo], dtype=float), atol=TOL,
                               rtol=0, equal_nan=True)
        # warm calls: served from the files, no RNG use
        st = np.random.get_state()[2]
        again = sim.get_fid_dists()
        assert list(again.keys()) == list(want_f.keys()) and np.random.get_state()[2] == st
        assert sim.get_metrics_dict() == json.load(open(sim.get_mcname() + "m"))


def test_mcdatasim_paper_scale_philox_metrics_only(workdir):
    """Paper scale (4 algorithms x 11 levels x 1000 controllers x 100 draws, N = 5), device draws, metrics only:
    RIM rows equal the mean infidelity of the (lazily fetched) tensors, level 0 (sigma = 0) equals the noiseless
    fidelity of every controller, NaN padding for the short algorithm."""
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    rng = np.random.default_rng(2)
    N, C, K = 5, 1000, 100
    def ctrls(m):
        return rand_ctrl(rng, m, N).tolist()
    le = {a: {"0.05": {"controller": ctrls(C)}} for a in ("ppo", "snob", "nmplus")}
    le["lbfgs"] = {str(N): {"controller": ctrls(825)}}
    os.makedirs("experiments/paper")
    json.dump(le, open(f"experiments/paper/ppo_spin_{N}_0-2_c_{C}", "w"))
    sim = mcmod.MCDataSim(experiment_name="paper", Nspin=N, inspin=0, outspin=2, bootreps=K, training_noise=0.05,
                          numcontrollers=C, verbose=False, rng_mode="philox", seed=9, cache_format="none")
    met = sim.get_metrics_dict()
    name = r'$W(.,\delta(x-1))$'
    for algo in ("ppo", "snob", "nmplus", "lbfgs"):
        rim = np.array(met[algo][name], dtype=float)
        assert rim.shape == (11, C)
        T = np.array(sim._metric_rows[algo][0])
        nvalid = 825 if algo == "lbfgs" else C
        assert np.isnan(T[:, nvalid:]).all() and np.isnan(rim[:, nvalid:]).all()
        assert np.abs(rim[:, :nvalid] - (1 - T[:, :nvalid]).mean(axis=2)).max() < 1e-13
        key = str(N) if algo == "lbfgs" else "0.05"
        x = np.array(le[algo][key]["controller"][:nvalid])
        assert np.abs(T[0, :nvalid, 0] - orc.fidelity_eigh(x, None, N, 0, 2)[:, 0]).max() < TOL
        assert (np.array(met[algo]["Q th. 0.95"], dtype=float)[:, nvalid:] == 0).all()      # Q of a NaN row: -0.0


# ----------------------------------------------------------------------------------------------------------------
# batched ARIM and the optimiser-side objective (8f-4), against the reference's own runs
# ----------------------------------------------------------------------------------------------------------------
def test_get_arims_matches_reference(workdir):
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    g = load_json("get_arims.json")
    sim = mcmod.MCDataSim(experiment_name="arims", Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"],
                          noises=np.array(g["noises"]), bootreps=g["bootreps"], numcontrollers=g["numcontrollers"],
                          verbose=False)
    cdict = json.loads(json.dumps(g["cdict"]))
    for run in g["runs"]:
        np.random.seed(run["seed"])
        arims, keys = sim.get_arims(run["algo"], nlvl="0.01", marker="nonstoch", cdict=cdict)
        assert abs(np.random.normal() - run["rng_after"]) < 1e-15
        assert keys == run["keys"] and np.abs(arims - np.array(run["arims"])).max() < TOL
        fname = os.path.join("experiments/arims", run["pickle"])
        assert np.array_equal(pickle.load(open(fname, "rb")), arims)
        again, none_keys = sim.get_arims(run["algo"], nlvl="0.01", marker="nonstoch", cdict=cdict)
        assert none_keys is None and np.array_equal(again, arims)
    assert cdict == g["cdict_after"]
    with pytest.raises(Exception, match="Unaccounted"):
        sim.get_arims("nope", cdict=cdict)


def test_fidelity_ss_av_matches_qnewton(be):
    """`noise.fidelity_ss_av` on the sets of `randHset_constructor` against `qnewton.LBFGS.fidelity_ss_av` of the
    unmodified reference (tests/golden/fidelity_ss_av.json): draw -> Hamiltonian mapping, seed-4 stream order,
    `reps` semantics, heisenberg_int through the public `HH`."""
    noise = importlib.import_module("code-robchar_amd.noise")
    g = load_json("fidelity_ss_av.json")
    for c in g["cases"]:
        n = c["Nspin"]
        nm = noise.structured_perturbation(Nspin=n, inspin=c["inspin"], outspin=c["outspin"], noise=c["sigma"])
        if c["heisenberg_int"]:
            nm.HH = nm.HH + np.diag(c["HH_diag"])
        train, test = nm.randHset_constructor(train_size=c["train_size"], test_size=c["test_size"])
        assert np.abs(train[:, :, 0] - np.array(c["train_diag"])).max() < 5e-16
        assert np.abs(train[:, 1:, 1] - np.array(c["train_sub"])).max() < 5e-16
        assert np.abs(test[:4, 1:, 1] - np.array(c["test_sub_head"])).max() < 5e-16
        x = np.array(c["controllers"])
        assert np.abs(nm.fidelity_ss_av(x, train, reps=c["reps"]) - np.array(c["av_train"])).max() < TOL
        assert np.abs(nm.fidelity_ss_av(x, train, reps=c["train_size"]) - np.array(c["av_train_all"])).max() < TOL
        assert np.abs(nm.fidelity_ss_av(x, test) - np.array(c["av_test"])).max() < TOL
        for i in range(len(x)):
            assert abs(nm.evaluate_noisy_fidelity(x[i]) - c["noiseless"][i]) < TOL


def test_mcdatasim_legacy_host_and_device_draws_agree(workdir):
    """legacy_draws='host' (NumPy draws, H2D) and 'device' (same stream on the GPU): same RNG position afterwards, same
    fidelities to 1e-13, both equal to the reference's seeded run."""
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    g = load_json("mcsim_run.json")
    run = g["runs"][0]
    res = {}
    for mode in ("host", "device"):
        _write_le(g, mode)
        np.random.seed(run["seed"])
        sim = mcmod.MCDataSim(experiment_name=mode, Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"],
                              noises=np.array(g["noises"]), bootreps=g["bootreps"], training_noise=run["training_noise"],
                              numcontrollers=g["numcontrollers"], filemarker=".le", verbose=False, legacy_draws=mode)
        fids = sim.get_fid_dists()
        assert abs(np.random.normal() - run["rng_after"]) < 1e-15
        res[mode] = {a: np.array(fids[a]) for a in fids}
    want = json.loads([t for f, t in run["files"].items() if f.endswith(".mc")][0])
    for a in want:
        assert np.allclose(res["host"][a], res["device"][a], atol=1e-13, rtol=0, equal_nan=True)
        assert np.allclose(res["device"][a], np.array(want[a], dtype=float), atol=TOL, rtol=0, equal_nan=True)


def test_scalar_api_lookahead_speed_and_exactness(be):
    """Reference-style loop over `evaluate_noisy_fidelity(x, True)`: K = 2000 single-sample calls are served by a handful
    of launches (look-ahead blocks), every value equals the oracle on the draws numpy actually produced, and the stream
    ends where 2000 x 3N scalar draws would leave it."""
    import time
    noise = importlib.import_module("code-robchar_amd.noise")
    N, K = 7, 2000
    rng = np.random.default_rng(17)
    x = rand_ctrl(rng, 1, N)[0]
    nm = noise.structured_perturbation(Nspin=N, inspin=0, outspin=6)
    np.random.seed(5)
    nm.rng(scale=0.05)
    nm.evaluate_noisy_fidelity(x, ham_noisy=True)                      # first call: library / kernel load
    t0 = time.perf_counter()
    got = np.array([nm.evaluate_noisy_fidelity(x, ham_noisy=True) for _ in range(K)])
    per_call = (time.perf_counter() - t0) / K
    after = np.random.normal()
    np.random.seed(5)
    np.random.normal(scale=0.05)
    draws = np.random.normal(scale=0.05, size=(1, K + 1, N, 3))
    assert after == np.random.normal()
    want = orc.fidelity_eigh(x[None, :], draws, N, 0, 6)[0, 1:]
    assert np.abs(got - want).max() < TOL
    print(f"scalar API: {per_call * 1e6:.1f} us per call (reference: ~100 us per evaluation at N = 7)")
    assert per_call < 60e-6


def test_mcdatasim_single_process_multi_device_mode(workdir):
    """`MCDataSim(devices=[...])`: all listed GPUs from ONE process through `rc_mc_metrics_sharded_f64` (here the one
    GPU of the box).  legacy: the reference's seeded run incl. RNG position; philox: identical to the one-GPU torch
    path (same stream offsets); metrics-only: the fidelities are not kept and say so."""
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    g = load_json("mcsim_run.json")
    run = g["runs"][0]
    want_m = json.loads([t for f, t in run["files"].items() if f.endswith(".mcm")][0])
    want_f = json.loads([t for f, t in run["files"].items() if f.endswith(".mc")][0])
    kw = dict(Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"], noises=np.array(g["noises"]), bootreps=g["bootreps"],
              training_noise=run["training_noise"], numcontrollers=g["numcontrollers"], filemarker=".le", verbose=False)
    _write_le(g, "md")
    np.random.seed(run["seed"])
    sim = mcmod.MCDataSim(experiment_name="md", devices=[0], **kw)
    met = sim.get_metrics_dict()
    fids = sim.get_fid_dists()
    assert abs(np.random.normal() - run["rng_after"]) < 1e-15
    for algo in want_f:
        assert np.allclose(np.array(fids[algo], dtype=float), np.array(want_f[algo], dtype=float), atol=TOL, rtol=0, equal_nan=True)
        for name in want_m[algo]:
            assert np.allclose(np.array(met[algo][name], dtype=float), np.array(want_m[algo][name], dtype=float), atol=TOL,
                               rtol=0, equal_nan=True), (algo, name)
    res = {}
    for tag, extra in (("one", {}), ("multi", {"devices": "all"})):
        _write_le(g, tag)
        sim = mcmod.MCDataSim(experiment_name=tag, rng_mode="philox", seed=5, **extra, **kw)
        res[tag] = (sim.get_metrics_dict(), {a: np.array(v) for a, v in sim.get_fid_dists().items()})
    for algo in res["one"][1]:
        assert np.array_equal(res["one"][1][algo], res["multi"][1][algo], equal_nan=True)
        for name in res["one"][0][algo]:
            assert np.array_equal(np.array(res["one"][0][algo][name]), np.array(res["multi"][0][algo][name]), equal_nan=True)
    _write_le(g, "mo")
    sim = mcmod.MCDataSim(experiment_name="mo", rng_mode="philox", seed=5, devices=[0], cache_format="none", **kw)
    met = sim.get_metrics_dict()
    assert met["ppo"].keys() == res["one"][0]["ppo"].keys()
    with pytest.raises(RuntimeError, match="not kept"):
        np.array(sim._metric_rows["ppo"][0])


@pytest.mark.parametrize("rng_mode", ["legacy", "philox"])
def test_mcdatasim_level_batching_is_transparent(tmp_path, monkeypatch, rng_mode):
    """`MCDataSim._run_algo` sends all sigma levels of an algorithm through ONE fidelity launch when one process makes the
    draws on the device; the level-by-level route (a buffer cap of zero) and the route with several generator calls per
    algorithm (a draw cap below one level pair) must give the same fidelities bit for bit and leave NumPy's stream at the
    same position (mcsim.py:422-460: noise outer, controller middle, draw inner, one burned draw per level)."""
    import importlib, json, os
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    monkeypatch.chdir(tmp_path)
    N, C, K = 5, 37, 24
    rng = np.random.default_rng(3)
    le = {}
    for a in ("ppo", "lbfgs"):
        x = np.empty((C if a == "ppo" else C - 5, N + 1))          # lbfgs: fewer controllers than asked for -> NaN rows
        x[:, :N] = rng.uniform(-10, 10, x[:, :N].shape)
        x[:, N] = rng.uniform(2, 30, x.shape[0])
        le[a] = {("%d" % N if a == "lbfgs" else "0.05"): {"controller": x.tolist()}}
    noises = np.linspace(0, 0.1, 6)
    results = []
    for tag, caps in (("batched", {}), ("level_by_level", {"_BATCH_LEVELS_MAX_BYTES": 0}),
                      ("several_generator_calls", {"_LEGACY_DEVICE_MAX_DRAWS": 2 * C * K * N * 3 + 7})):
        os.makedirs(f"experiments/{tag}")
        json.dump(le, open(f"experiments/{tag}/ppo_spin_{N}_0-2_c_{C}", "w"))
        for k, v in caps.items():
            monkeypatch.setattr(mcmod.MCDataSim, k, v)
        np.random.seed(11)
        sim = mcmod.MCDataSim(experiment_name=tag, Nspin=N, inspin=0, outspin=2, noises=noises, bootreps=K,
                              training_noise=0.05, numcontrollers=C, verbose=False, rng_mode=rng_mode, seed=5,
                              cache_format="json")
        fids = sim.get_fid_dists()
        results.append(({a: np.array(fids[a], dtype=float) for a in fids}, np.random.normal()))
        monkeypatch.undo()
        monkeypatch.chdir(tmp_path)
    ref, pos = results[0]
    assert set(ref) == {"ppo", "lbfgs"} and ref["ppo"].shape == (6, C, K) and np.isnan(ref["lbfgs"][:, C - 5:]).all()
    for got, p in results[1:]:
        assert p == pos
        for a in ref:
            assert np.array_equal(got[a], ref[a], equal_nan=True), a


def test_mcdatasim_philox_fused_route_equals_draw_tensor_route(tmp_path, monkeypatch):
    """`MCDataSim(rng_mode="philox")` now generates its draws inside the fidelity kernel (all sigma levels of an algorithm in one
    launch, one scale per tiled controller row); ROBCHAR_PHILOX_FUSED=0 takes the round-3 route (draw tensor + fidelity
    kernel, level batching).  Same stream elements, same arithmetic: the (L, C, K) tensors - NaN rows of a short controller
    list included -, the metric rows and NumPy's stream position (the burned draw per level, mcsim.py:425) are identical."""
    import importlib, json, os
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    monkeypatch.chdir(tmp_path)
    N, C, K = 7, 41, 130
    rng = np.random.default_rng(8)
    le = {}
    for a in ("ppo", "lbfgs"):
        x = np.empty((C if a == "ppo" else C - 6, N + 1))
        x[:, :N] = rng.uniform(-10, 10, x[:, :N].shape)
        x[:, N] = rng.uniform(2, 30, x.shape[0])
        le[a] = {("%d" % N if a == "lbfgs" else "0.05"): {"controller": x.tolist()}}
    noises = np.linspace(0, 0.1, 5)
    res = {}
    for tag, flag in (("fused", "1"), ("tensor", "0")):
        os.makedirs(f"experiments/{tag}")
        json.dump(le, open(f"experiments/{tag}/ppo_spin_{N}_0-3_c_{C}", "w"))
        monkeypatch.setenv("ROBCHAR_PHILOX_FUSED", flag)
        np.random.seed(21)
        sim = mcmod.MCDataSim(experiment_name=tag, Nspin=N, inspin=0, outspin=3, noises=noises, bootreps=K, training_noise=0.05,
                              numcontrollers=C, verbose=False, rng_mode="philox", seed=9, cache_format="json")
        fids = sim.get_fid_dists()
        met = sim.get_metrics_dict()
        res[tag] = ({a: np.array(fids[a], dtype=float) for a in fids}, met, np.random.normal())
    (f1, m1, p1), (f0, m0, p0) = res["fused"], res["tensor"]
    assert p1 == p0 and set(f1) == {"ppo", "lbfgs"}
    for a in f1:
        assert f1[a].shape == (5, C, K) and np.array_equal(f1[a], f0[a], equal_nan=True), a
        for k in m1[a]:
            assert np.array_equal(np.array(m1[a][k], dtype=float), np.array(m0[a][k], dtype=float), equal_nan=True), (a, k)


def test_mcdatasim_random_configurations_product_fuzz():
    """scripts/fuzz_mcdatasim.py, a short block of it: random `MCDataSim` calls - the GPU against the oracle-backed host route, the
    host-drawn against the device-continued legacy stream (generator state identical), the fused against the draw-tensor Philox route
    (bit for bit), the single-process multi-device route with 1 / 2 / 3 listed devices (identical)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SEED="3:4", NCFG="8", FUZZ_DEVICES="1", RC_ALLOW_DUPLICATE_DEVICES="1")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_mcdatasim.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "16 random MCDataSim configurations x 10 runs" in r.stdout, r.stdout[-2000:]
    print(r.stdout.strip().splitlines()[-6:])

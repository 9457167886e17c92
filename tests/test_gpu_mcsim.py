"""GPU end-to-end tests of the drop-in surface: `MCDataSim`, the noise model and the metric API running on
the HIP path, replayed against the seeded run of the unmodified reference (tests/golden/mcsim_run.json)."""
import importlib
import json
import os

import numpy as np
import pytest

from conftest import load_json
from oracle import robchar_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture
def workdir(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    os.mkdir("experiments")
    return tmp_path


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


def test_metric_api_on_gpu():
    rimm = importlib.import_module("code-robchar_amd.rim_metrics")
    g = load_json("metrics.json")
    for k, vec in g["vectors"].items():
        v = g["values"][k]
        a = np.array(vec, dtype=np.float64)
        assert abs(rimm.wd_from_ideal(a) - v["wd_from_ideal"]) < 1e-14
        assert np.array_equal(a, np.sort(np.array(vec, dtype=np.float64)))
        for p in (0, 1, 2, 3):
            assert abs(rimm.RIM_p(np.array(vec, dtype=np.float64), p) - v[f"RIM_{p}"]) < 1e-13
    slab = np.array(g["slab"], dtype=np.float64)
    tab = rimm.metric_table(slab)[""]
    for name, want in g["slab_metrics"].items():
        assert np.allclose(tab[name], want, atol=1e-14, rtol=0, equal_nan=True), name
    for name, fn in rimm.__metric_name_to_metric__.items():           # mcsim.py:144-183 under the reference's names, on the GPU
        assert np.allclose(list(fn(slab.copy())), g["slab_metrics"][name], atol=1e-14, rtol=0, equal_nan=True), name
    assert abs(rimm.Q(slab[0], 0.95) + g["slab_metrics"]["Q th. 0.95"][0]) < 1e-15
    # get_cdf (mcsim.py:42-47): GPU row sort + NumPy's own running sum = the reference's two arrays bit for bit, any length
    for row in (slab[0], np.random.default_rng(3).random(20001)):
        cdf, srt = rimm.get_cdf(row.copy())
        assert np.array_equal(srt, np.sort(row)) and np.array_equal(cdf, np.sort(row).cumsum() / np.sort(row).sum())
    with pytest.raises(TypeError):
        rimm.get_cdf(slab)                                            # the reference's 1-D guard (mcsim.py:35-39)
    # the reference's own unit-test identities (wd_sortof_fast_implementation.py:196-205)
    X = np.random.default_rng(0).normal(0.85, 0.8, size=10000).clip(min=0, max=1)
    mine = rimm.wd_from_ideal(X.copy())
    assert abs(np.sqrt(mine * mine + X.var()) - rimm.RIM_p(X, p=2)) < 1e-12
    from scipy.stats import wasserstein_distance
    assert abs(wasserstein_distance(X, np.ones_like(X)) - mine) < 1e-12


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


def test_directional_perturbation_on_gpu():
    """The `directional_perturbation` mirror end-to-end on the GPU against the reference's seeded run."""
    noise = importlib.import_module("code-robchar_amd.noise")
    g = load_json("directional.json")
    for case in g["cases"]:
        np.random.seed(case["seed"])
        nm = noise.directional_perturbation(Nspin=case["Nspin"], inspin=case["inspin"], outspin=case["outspin"],
                                            noise=case["sigma"])
        got = nm.fidelity_batch(np.array(case["controllers"]), case["K"], ham_noisy=True)
        assert abs(np.random.normal() - case["rng_after"]) < 1e-15
        assert np.abs(got - np.array(case["fid"])).max() < TOL
        x = np.array(case["controllers"][0])
        np.random.seed(case["seed"])
        nm2 = noise.directional_perturbation(Nspin=case["Nspin"], inspin=case["inspin"], outspin=case["outspin"],
                                             noise=case["sigma"])
        assert abs(nm2.evaluate_noisy_fidelity(x, ham_noisy=True) - case["fid"][0][0]) < TOL


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

"""CPU tests of the host-side mirror (MCDataSim / noise model / naming / metrics API) with the GPU calls
replaced by oracle-backed stand-ins (tests/stand_in.py).  What is under test here is everything AROUND the
kernels: file names, JSON layouts, cache policy, RNG consumption order, NaN padding, error behaviour -
replayed against the seeded run of the unmodified reference recorded in tests/golden/mcsim_run.json."""
import importlib
import json
import os

import numpy as np
import pytest

import stand_in
from conftest import load_json

mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
noise = importlib.import_module("code-robchar_amd.noise")
naming = importlib.import_module("code-robchar_amd.naming")
rimm = importlib.import_module("code-robchar_amd.rim_metrics")
libmod = importlib.import_module("code-robchar_amd._lib")


@pytest.fixture
def workdir(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    os.mkdir("experiments")
    return tmp_path


def _write_le(g):
    os.makedirs("experiments/golden", exist_ok=True)
    base = f"experiments/golden/ppo_spin_{g['Nspin']}_{g['inspin']}-{g['outspin']}_c_{g['numcontrollers']}"
    json.dump(g["le"], open(base + ".le", "w"))


def _sim(g, tn):
    return mcmod.MCDataSim(experiment_name="golden", Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"],
                           noises=np.array(g["noises"]), bootreps=g["bootreps"], training_noise=tn,
                           numcontrollers=g["numcontrollers"], filemarker=".le", verbose=False)


def test_seeded_run_matches_reference_files(workdir, monkeypatch):
    stand_in.install(monkeypatch)
    g = load_json("mcsim_run.json")
    _write_le(g)
    for run in g["runs"]:
        tn = run["training_noise"]
        np.random.seed(run["seed"])
        sim = _sim(g, tn)
        assert sim.algos == run["algos"]                     # empty 'snob' purged
        assert os.path.basename(sim.get_mcname()) == os.path.basename(run["mcname"])
        if tn is None:
            fids = sim.get_fid_dists(algoname="lbfgs")
        else:
            metrics = sim.get_metrics_dict()
            fids = sim.get_fid_dists()                       # cache hit, no RNG use
        assert abs(np.random.normal() - run["rng_after"]) < 1e-15, "RNG stream position differs"
        assert list(fids.keys()) == run["fids_keys"]
        written = sorted(f for f in os.listdir("experiments/golden") if ".mc" in f)
        assert written == sorted(run["files"].keys())
        for fname, text in run["files"].items():
            want = json.loads(text)
            got = json.load(open(os.path.join("experiments/golden", fname)))
            assert list(got.keys()) == list(want.keys())
            for algo in want:
                if fname.endswith(".mcm"):
                    assert list(got[algo].keys()) == list(want[algo].keys())
                    for name in want[algo]:
                        assert np.allclose(np.array(got[algo][name], dtype=float),
                                           np.array(want[algo][name], dtype=float), atol=1e-12, rtol=0,
                                           equal_nan=True), (algo, name)
                else:
                    w, h = np.array(want[algo], dtype=float), np.array(got[algo], dtype=float)
                    assert w.shape == h.shape
                    assert np.array_equal(np.isnan(w), np.isnan(h))
                    assert np.nanmax(np.abs(w - h)) < 1e-12
        for f in written:
            os.remove(os.path.join("experiments/golden", f))


def test_cache_policy_and_missing_algo(workdir, monkeypatch):
    stand_in.install(monkeypatch)
    g = load_json("mcsim_run.json")
    _write_le(g)
    sim = _sim(g, 0.05)
    np.random.seed(0)
    only = sim.get_fid_dists(algoname="ppo")
    assert list(only.keys()) == ["ppo"]
    st = np.random.get_state()[2]
    again = sim.get_fid_dists(algoname="ppo")                # served from the .mc file
    assert np.random.get_state()[2] == st and again.keys() == only.keys()
    with pytest.raises(Exception, match="unsuccessful"):
        sim.get_fid_dists(algoname="nmplus")                 # file holds ppo, which was not requested
    # .mcm is returned wholesale when present, whatever is asked
    json.dump({"sentinel": 1}, open(sim.get_mcname() + "m", "w"))
    assert sim.get_metrics_dict(algoname="ppo") == {"sentinel": 1}


@pytest.mark.parametrize("fmt", ["npy", "json"])
def test_resume_keeps_what_is_on_disk(workdir, monkeypatch, fmt):
    """Resuming an interrupted run: a NEW `MCDataSim` over an existing `.mc` file computes only the missing algorithm and
    APPENDS it.  In the npy format the loaded tensors are memory maps of the sidecars - re-dumping them would truncate
    the very file they map (this destroyed the cache before `McWriter.resume`); in both formats the bytes already on
    disk must stay untouched."""
    stand_in.install(monkeypatch)
    g = load_json("mcsim_run.json")
    _write_le(g)
    cio = importlib.import_module("code-robchar_amd.cache_io")

    def make():
        return mcmod.MCDataSim(experiment_name="golden", Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"],
                               noises=np.array(g["noises"]), bootreps=g["bootreps"], training_noise=0.05,
                               numcontrollers=g["numcontrollers"], filemarker=".le", verbose=False, cache_format=fmt)
    np.random.seed(3)
    first = make()
    ppo = np.array(first.get_fid_dists(algoname="ppo")["ppo"], dtype=float)
    path = first.get_mcname()
    side = path + ".ppo.npy"
    before = open(side if fmt == "npy" else path, "rb").read()
    if fmt == "json":
        open(path, "ab").write(b"\n")                      # a trailing newline must not break the append
    second = make()                                          # fresh object: no writer state, the file is all there is
    both = second.get_fid_dists()                            # every algorithm of the controller file; only ppo is cached
    order = ["ppo"] + [a for a in second.algos if a != "ppo"]          # cached first, then appended in `algos` order
    assert list(both) == order and len(order) >= 2
    new = order[1]
    assert np.array_equal(np.asarray(both["ppo"], dtype=float), ppo, equal_nan=True)
    after = open(side if fmt == "npy" else path, "rb").read()
    assert after[:len(before) - 1] == before[:-1]            # on-disk bytes of ppo untouched (json: up to the brace)
    third = cio.load_mc(path)
    assert list(third) == order
    assert np.array_equal(np.asarray(third["ppo"], dtype=float), ppo, equal_nan=True)
    assert np.array_equal(np.asarray(third[new], dtype=float), np.asarray(both[new], dtype=float), equal_nan=True)
    if fmt == "json":
        assert list(json.load(open(path))) == order            # still a file the reference's json.load reads
    # writer level: an identity-mismatched re-dump of memory-mapped tensors (the old crash) leaves the sidecars whole
    if fmt == "npy":
        w = cio.McWriter(path, json_max_values=1, cache_format="npy")
        w.dump(dict(third))
        again = cio.load_mc(path)
        assert np.array_equal(np.asarray(again["ppo"]), ppo, equal_nan=True) and list(again) == order


def test_missing_controller_file_is_flagged(workdir):
    sim = mcmod.MCDataSim(experiment_name="nothing_here", Nspin=4, outspin=3, numcontrollers=7, verbose=False)
    assert sim.controllers is None and sim.algos is None
    assert os.path.isdir("experiments/nothing_here")          # ExperimentNamer.home() side effect
    assert sim.get_controller_name == "experiments/nothing_here/ppo_spin_4_0-3_c_7"
    assert sim.get_mcname(0.1, np.linspace(0, 0.1, 11)).endswith(
        "_tn0.1_br_100_nlvl[0.   0.01 0.02 0.03 0.04 0.05 0.06 0.07 0.08 0.09 0.1 ].mc")


def test_experiment_namer_single_use(workdir):
    n = naming.ExperimentNamer(experiment_name="e1", Nspin=6, inspin=1, outspin=4, numcontrollers=3)
    assert n() == "experiments/e1/ppo_spin_6_1-4_c_3"
    assert n.home == "experiments/e1"                        # rebound to a str, as in the reference
    with pytest.raises(TypeError):
        n()


def test_noise_function_semantics():
    calls = []
    def gen(**kw):
        calls.append(dict(kw))
        return 0.5
    f = noise.noise_function(gen, scale=0.1)
    assert f() == 0.5 and f(scale=0.3, extra=1) == 0.5 and f() == 0.5
    assert calls == [{"scale": 0.1}, {"scale": 0.3, "extra": 1}, {"scale": 0.3, "extra": 1}]
    blk = f.draw_block((2, 3))                               # generator without size= -> element by element
    assert blk.shape == (2, 3) and "size" not in f.args


def test_noise_model_stream_equals_scalar_calls(monkeypatch):
    stand_in.install(monkeypatch)
    nm = noise.structured_perturbation(Nspin=5, inspin=0, outspin=2)
    np.random.seed(77)
    nm.rng(scale=0.05)
    got = nm.draw_samples(3, 4)
    np.random.seed(77)
    np.random.normal(scale=0.05)
    want = np.array([np.random.normal(scale=0.05) for _ in range(3 * 4 * 5 * 3)]).reshape(3, 4, 5, 3)
    assert np.array_equal(got, want)
    z = nm.perturbation()
    assert z.shape == (5, 5) and np.allclose(z, z.conj().T) and z[2, 0] == 0
    assert nm.HH[0, 1] == 1 and len(nm.CC) == 5 and nm.CC[3][3, 3] == 1
    ring = noise.structured_perturbation(Nspin=4, topo="ring")
    assert ring.HH[3, 0] == 1 and ring._static_terms()[2] is True
    nm.HH = nm.HH + np.diag([1.0, 0, 0, 0, 1.0])
    assert np.array_equal(nm._static_terms()[0], [1, 0, 0, 0, 1])
    nm.HH[0, 3] = 1
    with pytest.raises(NotImplementedError):
        nm._static_terms()


def test_get_rims_matches_reference(monkeypatch, workdir):
    stand_in.install(monkeypatch)
    g = load_json("get_rims.json")
    os.makedirs("experiments/r", exist_ok=True)
    sim = mcmod.MCDataSim(experiment_name="r", Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"],
                          noises=np.array(g["noises"]), bootreps=g["bootreps"], numcontrollers=1, verbose=False)
    np.random.seed(g["seed"])
    for cont, want in zip(g["controllers"], g["rims"]):
        assert np.abs(sim.get_rims(cont) - np.array(want)).max() < 1e-12
    assert abs(np.random.normal() - g["rng_after"]) < 1e-15


def test_metrics_api(monkeypatch):
    stand_in.install(monkeypatch)
    g = load_json("metrics.json")
    for k, vec in g["vectors"].items():
        v = g["values"][k]
        a = np.array(vec, dtype=np.float64)
        assert abs(rimm.wd_from_ideal(a) - v["wd_from_ideal"]) < 1e-14
        assert np.array_equal(a, np.sort(np.array(vec, dtype=np.float64)))      # sorted in place
        assert abs(rimm.wd_from_ideal_zero(list(vec)) - v["wd_from_ideal_zero"]) < 1e-14
        for p in (0, 1, 2, 3):
            assert abs(rimm.RIM_p(np.array(vec, dtype=np.float64), p) - v[f"RIM_{p}"]) < 1e-14
    assert abs(rimm.wd_from_ideal(0.76) - 0.24) < 1e-15                         # scalar input
    with pytest.raises(AssertionError):
        rimm.wd_from_ideal([0.2, 3.0])
    assert rimm.compute_dkw_error(0.05, 100) == pytest.approx(0.13581015157406195, abs=1e-16)
    slab = np.array(g["slab"], dtype=np.float64)
    tab = rimm.metric_table(slab)[""]
    for name, want in g["slab_metrics"].items():
        assert np.allclose(tab[name], want, atol=1e-14, rtol=0, equal_nan=True), name
    _check_metric_callables(g)


def _check_metric_callables(g):
    """mcsim.py:144-183 under the reference's names: the golden table IS `__metric_name_to_metric__` of the reference applied
    to this slab (tests/golden/make_golden.py: metrics)."""
    import types
    slab = np.array(g["slab"], dtype=np.float64)
    assert list(rimm.__metric_name_to_metric__) == list(g["slab_metrics"])                     # names and order of mcsim.py:178-183
    for name, fn in rimm.__metric_name_to_metric__.items():
        lazy = fn(slab.copy())
        assert isinstance(lazy, types.GeneratorType)                                           # nothing evaluated yet, like a map object
        assert np.allclose(list(lazy), g["slab_metrics"][name], atol=1e-14, rtol=0, equal_nan=True), name
    work = slab.copy()
    list(rimm.wd_from_ideal_fids(work))
    ok = ~np.isnan(slab).any(axis=1)
    assert np.array_equal(work[ok], np.sort(slab[ok], axis=1))                                 # rows sorted in place (wd...py:105)
    rows = [r.tolist() for r in slab[ok]]                                                      # lists of lists work like arrays
    assert np.allclose(list(rimm.std_fids(rows)), np.array(g["slab_metrics"]["std"])[ok], atol=1e-14)
    ragged = [slab[0], slab[1][:17], slab[4][:17]]                                             # rows of different lengths
    assert np.allclose(list(rimm.wc_fids(ragged)), [-slab[0].min(), -slab[1][:17].min(), -slab[4][:17].min()], atol=0)
    assert np.allclose(list(rimm.Q_fids(slab[ok], threshold=0.9)), [-(r >= 0.9).mean() for r in slab[ok]], atol=1e-15)
    assert np.allclose(list(rimm.Q_partial(qthres=0.98).Q_fids(slab[ok])), np.array(g["slab_metrics"]["Q th. 0.98"])[ok], atol=1e-15)
    assert rimm.Q(slab[0], 0.95) == pytest.approx(-g["slab_metrics"]["Q th. 0.95"][0], abs=1e-15)
    with pytest.raises(TypeError):
        rimm.Q(slab[0].tolist(), 0.95)                                                         # check_numpytype: 1-D arrays only
    with pytest.raises(AssertionError):
        list(rimm.wd_from_ideal_fids(np.array([[0.2, 3.0]])))                                  # check_fidtype's range guard


def test_product_path_fails_loudly_without_gpu():
    """No silent CPU fallback: on a box without a GPU the compute entry points raise."""
    be = importlib.import_module("code-robchar_amd.backend")
    if libmod.load().rc_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(libmod.RobCharHipError):
        be.mc_fidelity(np.ones((1, 6)), np.zeros((1, 1, 5, 3)), 5, 0, 2)
    with pytest.raises(libmod.RobCharHipError):
        be.reduce_metrics(np.ones((1, 4)))
    assert "oracle" not in open(be.__file__).read().replace("test infrastructure", "")


def test_merge_tools_and_cli(workdir, monkeypatch, capsys):
    """merge_controller_files / merge_mcdata (mcsim.py:594-649) and the CLI built on parse.py's MC flags."""
    stand_in.install(monkeypatch)
    g = load_json("mcsim_run.json")
    _write_le(g)
    os.makedirs("experiments/other", exist_ok=True)
    base = f"ppo_spin_{g['Nspin']}_{g['inspin']}-{g['outspin']}_c_{g['numcontrollers']}.le"
    other = {"nmplus": {"0.1": {"controller": g["le"]["nmplus"]["0.0"]["controller"]}},
             "brandnew": {"0.0": {"controller": g["le"]["ppo"]["0.0"]["controller"]}}}
    json.dump(other, open("experiments/other/" + base, "w"))
    sim = _sim(g, 0.05)
    sim.merge_controller_files("other")
    merged = json.load(open("experiments/golden/" + base))
    assert "brandnew" in merged and "0.1" in merged["nmplus"] and "0.0" in merged["nmplus"]
    assert merged["lbfgs"] == g["le"]["lbfgs"]
    with pytest.raises(naming.DirectoryDoesNotExistError):
        sim.get_path("does_not_exist")
    # caches: compute here for ppo, there for nmplus, merge
    np.random.seed(1)
    _write_le(g)                                              # restore the original controller file
    sim = _sim(g, 0.05)
    sim.algos = ["ppo"]
    sim.get_metrics_dict()
    here = sim.get_mcname()
    os.makedirs("experiments/other2", exist_ok=True)
    json.dump(g["le"], open("experiments/other2/" + base, "w"))
    sim2 = mcmod.MCDataSim(experiment_name="other2", Nspin=g["Nspin"], inspin=g["inspin"], outspin=g["outspin"],
                           noises=np.array(g["noises"]), bootreps=g["bootreps"], training_noise=0.05,
                           numcontrollers=g["numcontrollers"], filemarker=".le", verbose=False)
    sim2.algos = ["nmplus"]
    sim2.get_metrics_dict()
    sim.merge_mcdata("other2")
    assert set(json.load(open(here)).keys()) == {"ppo", "nmplus"}
    assert set(json.load(open(here + "m")).keys()) == {"ppo", "nmplus"}
    assert isinstance(json.load(open(here))["nmplus"], list)          # fidelities stayed in the .mc file
    # CLI
    cli = importlib.import_module("code-robchar_amd.cli")
    for f in os.listdir("experiments/golden"):
        if ".mc" in f:
            os.remove("experiments/golden/" + f)
    rc = cli.main(["--exp_name", "golden", "--nspin", str(g["Nspin"]), "--outspin", str(g["outspin"]), "--bootreps",
                   "5", "--training_noise", "0.05", "--mc_max_noise", "0.1", "--mc_noise_res", "3", "--numcontrollers",
                   str(g["numcontrollers"]), "--filemarker", ".le", "--seed", "1234"])
    out = capsys.readouterr().out
    assert rc == 0 and "mean RIM per sigma_sim" in out
    want = json.loads(g["runs"][0]["files"][[k for k in g["runs"][0]["files"] if k.endswith(".mc")][0]])
    got = json.load(open(here))
    for algo in want:
        assert np.allclose(np.array(got[algo], dtype=float), np.array(want[algo], dtype=float), atol=1e-12, rtol=0,
                           equal_nan=True)


def test_directional_mirror_rng_and_layout(monkeypatch):
    """`directional_perturbation` mirror: same RNG consumption (randint + normal(size=2) per sample, sticky size)
    and same fidelities as the reference's seeded run."""
    stand_in.install(monkeypatch)
    g = load_json("directional.json")
    for case in g["cases"][:2]:
        n = case["Nspin"]
        np.random.seed(case["seed"])
        nm = noise.directional_perturbation(Nspin=n, inspin=case["inspin"], outspin=case["outspin"], noise=case["sigma"])
        assert [list(d) for d in nm.directions] == case["directions"]
        got = nm.fidelity_batch(np.array(case["controllers"]), case["K"], ham_noisy=True, draws="host")   # (the host mirror)
        assert abs(np.random.normal() - case["rng_after"]) < 1e-15
        assert np.abs(got - np.array(case["fid"])).max() < 1e-12
        assert nm.rng.args.get("size") == 2                 # sticky, as in the reference
        z = nm.perturbation()
        assert z.shape == (n, n) and np.count_nonzero(z) in (1, 2)


def test_scalar_api_lookahead_keeps_the_reference_stream(monkeypatch):
    """`evaluate_noisy_fidelity(x, True)` looks ahead (one launch for a block of samples) but leaves numpy's generator
    exactly where the reference would be after EVERY call; interleaved use of the stream by anybody else, another
    controller or another sigma drops the block.  Replayed against straightforward one-sample-at-a-time evaluation."""
    from oracle import robchar_oracle as orc
    be = stand_in.install(monkeypatch)
    calls = []
    real = be.mc_fidelity
    monkeypatch.setattr(be, "mc_fidelity", lambda *a, **k: (calls.append(a[1].shape), real(*a, **k))[1])
    N, a, b = 5, 0, 2
    rng = np.random.default_rng(3)
    x1 = np.concatenate([rng.uniform(-10, 10, N), [7.3]])
    x2 = np.concatenate([rng.uniform(-10, 10, N), [11.0]])

    def reference_like(script):
        """The reference's consumption: per evaluation 3N scalar draws, evaluated one at a time by the oracle."""
        out = []
        for op in script:
            if op[0] == "eval":
                g = np.random.normal(scale=op[2], size=(1, 1, N, 3))
                out.append(orc.fidelity_eigh(op[1][None, :], g, N, a, b)[0, 0])
            elif op[0] == "burn":
                out.append(np.random.normal(scale=op[1]))
            else:
                out.append(np.random.random())
        return out

    script = ([("burn", 0.05)] + [("eval", x1, 0.05)] * 30 + [("other",)] + [("eval", x1, 0.05)] * 3
              + [("eval", x2, 0.05)] * 5 + [("burn", 0.1)] + [("eval", x2, 0.1)] * 40)
    np.random.seed(11)
    want = reference_like(script)
    want_state = np.random.get_state()
    nm = noise.structured_perturbation(Nspin=N, inspin=a, outspin=b)
    np.random.seed(11)
    got = []
    for op in script:
        if op[0] == "eval":
            got.append(nm.evaluate_noisy_fidelity(op[1], ham_noisy=True))
        elif op[0] == "burn":
            got.append(nm.rng(scale=op[1]))
        else:
            got.append(np.random.random())
    st = np.random.get_state()
    assert np.array_equal(st[1], want_state[1]) and st[2:] == want_state[2:]
    assert np.abs(np.array(got) - np.array(want)).max() < 1e-12
    assert len(calls) < 20 and max(c[1] for c in calls) >= 32           # 78 evaluations, a handful of launches
    # noiseless calls and foreign generators never look ahead
    assert abs(nm.evaluate_noisy_fidelity(x1) - orc.fidelity_eigh(x1[None, :], None, N, a, b)[0, 0]) < 1e-12
    nm2 = noise.structured_perturbation(Nspin=N, inspin=a, outspin=b, rng=noise.noise_function(lambda **k: 0.01))
    assert not nm2._lookahead_usable() and 0 <= nm2.evaluate_noisy_fidelity(x1, ham_noisy=True) <= 1


def test_scalar_api_lookahead_directional_model(monkeypatch):
    """The same look-ahead for `directional_perturbation` (round 4; round 3 paid one launch + sync per sample there): per call
    `np.random.randint(0, len(directions))` then `rng(size=2)` are consumed from the live stream exactly as the reference does
    (noise_model.py:183-189), the block behind it comes from the bit-identical host emulation of that consumption, numpy's state
    is the reference's after every call; foreign draws, another controller, a burned (size-2: sticky) draw drop the block."""
    from oracle import robchar_oracle as orc
    be = stand_in.install(monkeypatch)
    calls = []
    real = be.mc_fidelity
    monkeypatch.setattr(be, "mc_fidelity", lambda *a, **k: (calls.append(a[1].shape), real(*a, **k))[1])
    N, a, b = 5, 0, 4
    rng = np.random.default_rng(4)
    x1 = np.concatenate([rng.uniform(-10, 10, N), [6.1]])
    x2 = np.concatenate([rng.uniform(-10, 10, N), [9.0]])
    ndir = len(orc.directional_directions(N))

    def reference_like(script):
        out = []
        for op in script:
            if op[0] == "eval":
                idx = np.random.randint(low=0, high=ndir)
                ab = np.random.normal(scale=op[2], size=2)
                g, im = orc.directional_to_layout(N, idx, ab[0], ab[1])
                out.append(orc.fidelity_expm_loop(op[1][None, :], g[None, None], N, a, b, diag_imag=im[None, None])[0, 0])
            elif op[0] == "burn":
                out.append(float(np.random.normal(scale=op[1], size=2)[0]))      # `size=2` is sticky on the generator by then
            else:
                out.append(np.random.random())
        return out

    script = ([("eval", x1, 0.05)] * 30 + [("other",)] + [("eval", x1, 0.05)] * 3 + [("eval", x2, 0.05)] * 5 + [("burn", 0.1)]
              + [("eval", x2, 0.1)] * 40)
    np.random.seed(12)
    want = reference_like(script)
    want_state = np.random.get_state()
    nm = noise.directional_perturbation(Nspin=N, inspin=a, outspin=b, noise=0.05)
    np.random.seed(12)
    got = []
    for op in script:
        if op[0] == "eval":
            got.append(nm.evaluate_noisy_fidelity(op[1], ham_noisy=True))
        elif op[0] == "burn":
            got.append(float(nm.rng(scale=op[1])[0]))
        else:
            got.append(np.random.random())
    st = np.random.get_state()
    assert np.array_equal(st[1], want_state[1]) and st[2:] == want_state[2:]
    assert np.abs(np.array(got) - np.array(want)).max() < 1e-10
    assert len(calls) < 20 and max(c[1] for c in calls) >= 32           # 78 evaluations, a handful of launches


def test_native_json_cache_writer(tmp_path):
    """`cache_io` / `rc_json_*` (host code of the C ABI library): the text parses back to the identical doubles with
    Python's own `json`, has the bracket structure `json.dumps` produces for every shape (degenerate ones included),
    keeps integral values floats, writes NaN / Infinity like `json.dump`; multi-round multi-thread file output lands at
    the right offsets; `McWriter` appends algorithm by algorithm and switches to npy sidecars above the threshold."""
    import re
    cio = importlib.import_module("code-robchar_amd.cache_io")
    rng = np.random.default_rng(0)
    for shape in [(5,), (3, 4), (2, 3, 4), (11, 100, 7), (1, 1, 1), (2, 0), (0, 3), (3, 0, 2), (4, 1), (2, 2, 2, 2), (3, 2, 0)]:
        a = rng.random(shape)
        if a.size > 2:
            a.flat[0], a.flat[1], a.flat[2] = np.nan, 1.0, -0.0
        txt = bytes(cio.encode_array(a)).decode()
        assert np.array_equal(np.array(json.loads(txt), dtype=float).reshape(shape), a, equal_nan=True), shape
        assert re.sub(r"[^\[\], ]", "", txt) == re.sub(r"[^\[\], ]", "", json.dumps(a.tolist())), shape
    vals = np.array([1e-5, 1e-4, 1e16, 5e-324, 1.7976931348623157e308, np.inf, -np.inf, 123456789.0, 1e22, 0.1 + 0.2, -0.0])
    back = json.loads(bytes(cio.encode_array(vals)))
    assert all(type(b) is float for b in back) and np.array_equal(np.array(back), vals) and np.signbit(back[-1])
    big = rng.random((7, 1500, 100))                               # 1.05e6 values: 17 blocks, several rounds of threads
    big[3, 77] = np.nan
    path = str(tmp_path / "big.json")
    cio.write_json({"ppo": big, "nested": {"a": big[0], "n": 3}}, path)
    got = json.load(open(path))
    assert np.array_equal(np.array(got["ppo"], dtype=float), big, equal_nan=True)
    assert np.array_equal(np.array(got["nested"]["a"]), big[0]) and got["nested"]["n"] == 3
    # a `.mcm`-shaped dict: many mid-size leaves, encoded side by side (one thread per leaf) and written in order;
    # one array object appearing twice, a NaN row, a small and a 1-d leaf in between
    shared = rng.random((11, 700))
    mcm = {a: {f"m{j}": (shared if j == 3 else rng.random((11, 700))) for j in range(6)} for a in ("ppo", "snob", "lbfgs")}
    mcm["ppo"]["m0"][2, :] = np.nan
    mcm["snob"]["small"] = rng.random((3, 5))
    mcm["lbfgs"]["vec"] = rng.random(5000)
    path = str(tmp_path / "x.mcm")
    cio.write_json(mcm, path)
    got = json.load(open(path))
    assert list(got) == list(mcm) and all(list(got[a]) == list(mcm[a]) for a in mcm)
    for a in mcm:
        for k, v in mcm[a].items():
            assert np.array_equal(np.array(got[a][k], dtype=float), v, equal_nan=True), (a, k)
    mc = str(tmp_path / "x.mc")
    w = cio.McWriter(mc, json_max_values=10 ** 7)
    sim = {"ppo": big}
    w.dump(sim)
    sim["snob"] = rng.random((2, 3, 4))
    w.dump(sim)
    sim["lbfgs"] = rng.random((2, 3, 4)).tolist()
    w.dump(sim)
    got = cio.load_mc(mc)
    assert list(got) == ["ppo", "snob", "lbfgs"] and got["lbfgs"] == sim["lbfgs"]
    assert np.array_equal(np.array(got["ppo"], dtype=float), big, equal_nan=True)
    w2 = cio.McWriter(mc, json_max_values=10)
    w2.dump(sim)
    got = cio.load_mc(mc)
    assert json.load(open(mc))["__robchar_npy__"] == 1 and np.array_equal(got["snob"], sim["snob"])
    assert np.array_equal(np.asarray(got["ppo"]), big, equal_nan=True)


def test_native_json_text_equals_json_dumps_byte_for_byte():
    """The native encoder writes exactly what `json.dump` writes (mcsim.py:457-459: the cache files ARE the boundary):
    `float.__repr__`'s notation rule - positional while the decimal point sits at -4 < decpt <= 16, exponent notation
    outside - not std::to_chars' "whichever is shorter" (which gave "1e-04" for 0.0001 and 21 positional digits beyond 1e16;
    the values round-tripped, the text differed).  10^5 random bit patterns over the full exponent range, uniform [0, 1)
    fidelities, scaled values, and the edges of both notation switches."""
    cio = importlib.import_module("code-robchar_amd.cache_io")
    rng = np.random.default_rng(20220714)
    a = rng.integers(0, 2 ** 64, 120000, dtype=np.uint64).view(np.float64)
    a = a[np.isfinite(a)][:100000]
    edge = [0.0, -0.0, 1.0, -1.0, 1e16, 1e15, 9999999999999998.0, 1.2345678901234567e16, 123456789012345680.0, 1e-4, 1e-5,
            0.0001234, 1.5e-5, 1e22, 1e23, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 0.1, 0.5, 1e-7, 123.0,
            0.001, 0.00012345678901234567, 1e17, 1.5e300, -872793927382035857408.0, np.nan, np.inf, -np.inf]
    for pw in range(-30, 31):
        edge += [10.0 ** pw, -(10.0 ** pw), 3 * 10.0 ** pw, 1.25 * 10.0 ** pw, np.nextafter(10.0 ** pw, 0), np.nextafter(10.0 ** pw, np.inf)]
    vals = np.concatenate([a, np.array(edge), rng.random(50000), rng.uniform(-100, 100, 20000) * 10.0 ** rng.integers(-20, 20, 20000)])
    assert bytes(cio.encode_array(vals)).decode() == json.dumps(vals.tolist())
    cube = vals[:60000].reshape(100, 20, 30)                       # nested lists: brackets and separators as json.dumps lays them
    assert bytes(cio.encode_array(cube)).decode() == json.dumps(cube.tolist())


def test_directional_host_emulation_equals_numpy(monkeypatch):
    """`rc_directional_draws_legacy` (host emulation of the legacy stream for the interleaved randint / normal(size=2)
    consumption of noise_model.py:183-189) against NumPy itself: indices, normals and generator state bit-identical from
    arbitrary stream positions; `draw_samples` maps them to the kernel layout like the sample-by-sample loop."""
    stand_in.install(monkeypatch)
    for n_spin, seed, count in ((4, 1, 1), (5, 2, 17), (7, 3, 5000), (10, 4, 777)):
        nm = noise.directional_perturbation(Nspin=n_spin, inspin=0, outspin=n_spin - 1, noise=0.07)
        np.random.seed(seed)
        np.random.standard_normal(seed)                     # odd seeds leave a cached normal behind
        idx, ab = nm._draw_indices(count)
        st = np.random.get_state()
        np.random.seed(seed)
        np.random.standard_normal(seed)
        want_idx = np.empty(count, dtype=np.int64)
        want_ab = np.empty((count, 2))
        for i in range(count):
            want_idx[i] = np.random.randint(low=0, high=len(nm.directions))
            want_ab[i] = np.random.normal(scale=0.07, size=2)
        st2 = np.random.get_state()
        assert np.array_equal(idx, want_idx) and np.array_equal(ab, want_ab)
        assert np.array_equal(st[1], st2[1]) and st[2:] == st2[2:]
        assert nm.rng.args.get("size") == 2
    # layout: every sample perturbs exactly one element pair
    nm = noise.directional_perturbation(Nspin=6, inspin=0, outspin=3, noise=0.1)
    np.random.seed(9)
    draws, imag = nm.draw_samples(3, 50)
    nz = (draws != 0).reshape(150, -1).sum(axis=1) + (imag != 0).reshape(150, -1).sum(axis=1)
    assert ((nz == 2) | (nz == 1)).all() and (imag != 0).any() and (draws[..., 2] != 0).any()


def test_philox_fused_routing_rule():
    """Where `MCDataSim` (and the sharded C entries, which apply the same rule) take the fused Philox fidelity kernel: chain,
    eigenvalue-only kernels, and only the sizes at which it is the faster of two bit-identical routes
    (profiles/r04_philox_fused_sweep.txt)."""
    be = importlib.import_module("code-robchar_amd.backend")
    assert be.philox_fused_supported(7) and be.philox_fused_supported(16) and not be.philox_fused_supported(17)
    assert not be.philox_fused_supported(7, ring=True) and not be.philox_fused_supported(7, kernel="tridiag_ql")
    for n in range(2, 14):
        assert be.philox_fused_pays(n, 0, n - 1) and be.philox_fused_pays(n, n // 2, 0)
    assert be.philox_fused_pays(14, 0, 13) and be.philox_fused_pays(14, 13, 0) and not be.philox_fused_pays(14, 0, 7)
    assert not be.philox_fused_pays(15, 0, 14) and not be.philox_fused_pays(16, 0, 15) and not be.philox_fused_pays(16, 4, 9)

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
    assert "static" in rf["traffic_source"] and "static" in d["fp64_valu"]["source"]      # labelled, not "measured"
    # the appended strong-scaling run of BASELINE config 4 and the product-API timings travel in the same line
    assert d["config"]["library"]["build_flags"] == 0 and d["config"]["library"]["abi"] >= 6
    assert d["config"]["library"]["legacy_normals_bit_identical_to_numpy"] is True
    assert d["extras_failed"] == [] and d["config"]["rccl"]["world"] == 1 and len(d["config"]["rccl"]["devices"]) == 1
    assert 0 < d["also"]["cold_20_steps_kernel_ms"]["kernel_ms"] < 1.0
    sh = d["also"]["shipped_lbfgs_controllers"]          # SURVEY.md 8(d)'s realistic variant: the reference's shipped controllers
    assert sh["max_abs_err_vs_oracle_2pct"] < 1e-10 and sh["max_abs_err_noiseless_vs_reference_best_fid"] < 1e-10
    assert 0 < sh["kernel_ms"] < 1.0 and 0.5 < sh["mean_fidelity"] <= 1.0
    # round 5: every parity figure says what it was measured ON - the headline's uniform random controllers are localised
    # (median fidelity ~1e-7: disclosed, not hidden), the appended legs run each configuration's timed kernel on fidelities of
    # O(1) with a relative bound beside the absolute one
    for k in ("median_fidelity", "frac_F_gt_1e-3", "max_rel_err_F_gt_1e-3"):
        assert k in d["check"] and k in d["also"]["config4_strong"]["check"] and k in sh, k
    assert d["check"]["median_fidelity"] < 1e-3                       # SURVEY 8(d)'s synthetic controllers: Anderson-localised
    legs = [sh] + [d["also"]["delocalised"][f"config{c}"] for c in (2, 4, 5)]
    for lg in legs:
        assert lg["median_fidelity"] > 0.1 and lg["frac_F_gt_1e-3"] > 0.9 and lg["compared_samples_F_gt_1e-3"] > 10000, lg
        assert lg["max_abs_err_vs_oracle_2pct"] < 1e-10 and lg["max_rel_err_F_gt_1e-3"] < 1e-9, lg
        assert 0 < lg["kernel_ms"] < 1.0 and 0 < lg["roofline_frac"] < 1
    # round 5: the headline kernel at 20 x the benchmark's launch size - what a launch boundary (fill + drain of 3.8 rounds of
    # waves) costs the 1e6-evaluation step: measured 12 % faster per evaluation than the headline (the bound here only says
    # "same kernel, same order of magnitude": a timing relation between two legs of one run is not a correctness property)
    ls = d["also"]["launch_size"]
    assert ls["max_abs_err_vs_oracle"] < 1e-10 and ls["compared_samples"] >= 100
    assert 0.5 * rf["kernel_ms"] < ls["kernel_ms_per_1e6_evals"] < 1.25 * rf["kernel_ms"] and 0 < ls["roofline_frac"] < 1
    c4 = d["also"]["config4_strong"]
    assert c4["scaling"] == "strong" and c4["n_gpus"] == 1 and c4["check"]["max_abs_err_vs_oracle"] < 1e-10
    assert abs(c4["value"] - 1e8 / (c4["ms_per_step"] * 1e-3)) / c4["value"] < 1e-3
    assert d["config"]["config4_strong_evals_per_s"] == c4["value"]
    assert d["config"]["end_to_end_wall_s"]["c4_level_api"] == d["end_to_end"]["c4_level_api"]["wall_s"]
    tail = r.stdout.strip().splitlines()[-1][-2000:]
    assert '"end_to_end"' in tail and '"paper_philox_metrics_only"' in tail          # fits the tail a driver record keeps
    e2e = d["end_to_end"]
    for leg in ("paper_philox_metrics_only", "paper_philox_json_cache", "paper_philox_npy_cache", "paper_legacy_json_cache",
                "c4_level_api", "arim_scan_legacy"):
        assert e2e[leg]["wall_s"] > 0 and e2e[leg]["evals_per_s"] > 0, leg
    assert e2e["paper_philox_metrics_only"]["evals"] == 4 * 11 * 1000 * 100 and e2e["c4_level_api"]["evals"] == 10**8


def test_bench_config4_explicit():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--steps", "6", "--warmup", "2",
                        "--no-end-to-end"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["scaling"] == "strong" and d["steps"] == 6 and "config 4" in d["config"]["workload"]
    assert d["config"]["evals_per_step"] == 10**8 and d["roofline"]["evals_per_launch"] == 10**8
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["check"]["max_abs_err_vs_oracle"] < 1e-10 and d["check"]["rim_err"] < 1e-10


def test_bench_config4_with_draws_generated_in_the_kernel():
    """`--config 40`: BASELINE config 4 with the counter-based draws generated inside the fidelity kernel in every step - the
    same stream elements as `--config 4` generates once, so the LAST step's metric table equals config 4's bit for bit."""
    out = {}
    for cfg in ("40", "4"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "3", "--warmup", "1",
                            "--no-end-to-end", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        out[cfg] = _json_line(r.stdout)
    d = out["40"]
    assert d["scaling"] == "strong" and d["config"]["evals_per_step"] == 10**8 and "inside the fidelity kernel" in d["config"]["workload"]
    assert "philox_kernel" in d["roofline"]["kernel"] and d["roofline"]["bytes_per_eval"] == 8
    assert d["check"]["max_abs_err_vs_oracle"] < 1e-10 and d["check"]["rim_err"] < 1e-10
    assert d["check"]["metric_table_sha256"] == out["4"]["check"]["metric_table_sha256"]
    assert d["ms_per_step"] < 1.6 * out["4"]["ms_per_step"]            # a whole level for less than 1.6 x its second half


@pytest.mark.parametrize("config,evals,kernel", [(2, 10**6, "<5, 2>"), (5, 10**6, "<10, 2>")])
def test_bench_other_baseline_configs(config, evals, kernel):
    """BASELINE configs 2 (N = 5) and 5 (N = 10 XXZ) as bench modes: same contract, CPU baseline on the whole workload."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--steps", "40", "--warmup", "5",
                        "--no-end-to-end"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert f"config {config}" in d["config"]["workload"] and d["config"]["evals_per_step"] == evals
    assert kernel in d["roofline"]["kernel"] and d["roofline"]["bytes_per_eval"] == 24 * (5 if config == 2 else 10) + 8
    assert d["check"]["max_abs_err_vs_cpu_baseline_all_1e6"] < 1e-10 and d["check"]["rim_err"] < 1e-10
    assert d["cpu_baseline"]["value"] > 0 and d["roofline"]["traffic"] is None


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
    # strong-scaling config 4: rank r owns 500 controllers and its own slice of the Philox stream; metric rows gathered
    c4 = d["also"]["config4_strong"]
    assert c4["n_gpus"] == 2 and c4["scaling"] == "strong" and c4["check"]["gather_ok"]
    assert c4["evals_per_step"] == 10**8 and c4["evals_per_launch"] == 5 * 10**7
    # the sharded product API (MCDataSim under the process group): metric rows all-gathered, only rank 0 writes
    assert d["end_to_end"]["c4_level_api"]["evals"] == 10**8 and d["end_to_end"]["paper_philox_json_cache"]["wall_s"] > 0


def test_bench_single_rank_rccl_branch():
    """RCCL itself on the one-GPU box: a ONE-rank `nccl` process group (RCCL refuses two ranks on one device), every
    collective of the benchmark and of the sharded `MCDataSim` executed on it - communicator set-up with `device_id`,
    `all_gather_into_tensor` on the high-priority side stream, scatter / broadcast of the legacy-stream state."""
    env = dict(os.environ, ROBCHAR_BENCH_FORCE_PG="1", ROBCHAR_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29573",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "35", "--warmup", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _json_line(r.stdout)
    assert d["config"]["collective"] == "rccl all_gather_into_tensor" and d["check"]["gather_ok"]
    assert d["config"]["rccl"]["backend"] == "nccl" and d["config"]["rccl"]["world"] == 1      # read from the communicator
    assert d["also"]["config4_strong"]["collective"] == "rccl all_gather_into_tensor"
    assert d["also"]["config4_strong"]["check"]["gather_ok"]
    assert d["end_to_end"]["paper_legacy_json_cache"]["wall_s"] > 0 and d["end_to_end"]["c4_level_api"]["evals"] == 10**8


def test_bench_extras_watchdog_keeps_the_headline():
    """The extras (appended config-4 run, product-API legs) run collectives of their own under N > 1; a rank lost in
    them must not cost the headline.  With the deadline set to (almost) nothing the watchdog fires during the extras:
    the ONE line is still printed - headline intact, the unfinished extras marked, the leg in flight named - and the
    exit code is the distinct EXIT_EXTRAS (3), not 0: a hang in the extras is visible to whoever reads the rc."""
    env = dict(os.environ, ROBCHAR_BENCH_EXTRAS_TIMEOUT_S="0.05")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 3, (r.stdout + r.stderr)[-3000:]
    d = _json_line(r.stdout)
    assert d["steps"] == 20 and d["value"] > 0 and d["check"]["max_abs_err_vs_oracle"] < 1e-10
    assert "watchdog" in d["end_to_end"]["error"] and "leg in flight" in d["end_to_end"]["error"]
    assert d["extras_failed"] and d["extras_failed"][0].startswith("watchdog:")
    assert "watchdog fired" in r.stderr


def test_bench_self_launch_two_ranks():
    """`python3 bench.py --gpus 2 --steps 20 --warmup 5` exactly as a driver would type it - no external launcher: the
    script starts its two ranks itself (gloo rehearsal backend on the one-GPU box, RCCL refuses two ranks on one device).
    The line proves what the communicator saw (`config.rccl`) and carries both exchange variants of config 4."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["ROBCHAR_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=1100, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["extras_failed"] == [] and d["check"]["gather_ok"] and d["check"]["max_abs_err_vs_oracle"] < 1e-10
    rc = d["config"]["rccl"]
    assert rc["world"] == 2 and rc["backend"] == "gloo" and len(rc["devices"]) == 2
    assert sorted(x["rank"] for x in rc["devices"]) == [0, 1] and rc["distinct_devices"] == 1      # one GPU on this box
    assert "self-launched" in d["config"]["launcher"]
    for legname in ("config4_strong", "config4_strong_gather_fid"):
        c4 = d["also"][legname]
        assert c4["n_gpus"] == 2 and c4["check"]["gather_ok"] and c4["check"]["max_abs_err_vs_oracle"] < 1e-10, legname
        assert c4["evals_per_step"] == 10**8 and c4["evals_per_launch"] == 5 * 10**7
    assert d["also"]["cold_20_steps_kernel_ms"]["kernel_ms"] > 0


def test_bench_self_launch_two_gpus_rccl():
    """The same command on a box with at least TWO GPUs, RCCL over xGMI (skipped on the one-GPU boxes of this pool): two
    ranks on two distinct devices, `config.rccl` read from the communicator, both exchange variants of config 4."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "ROBCHAR_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=1100, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _json_line(r.stdout)
    rc = d["config"]["rccl"]
    assert d["n_gpus"] == 2 and rc["world"] == 2 and rc["backend"] == "nccl" and rc["distinct_devices"] == 2
    assert d["config"]["collective"] == "rccl all_gather_into_tensor" and d["check"]["gather_ok"] and d["extras_failed"] == []
    for legname in ("config4_strong", "config4_strong_gather_fid"):
        assert d["also"][legname]["check"]["gather_ok"] and d["also"][legname]["n_gpus"] == 2, legname


def _self_launched(extra_args, nranks, timeout=1100):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["ROBCHAR_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + (["--gpus", str(nranks)] if nranks > 1 else []) + extra_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return _json_line(r.stdout)


@pytest.mark.parametrize("config,steps,total", [(5, 35, 100), (30, 35, 100), (4, 3, 1000), (40, 3, 1000)])
def test_bench_three_ranks_ragged_equals_one_rank(config, steps, total):
    """Strong scaling over THREE ranks (34/33/33 and 334/333/333 controllers: the ragged branch of the reduction stage -
    padded metric tables, non-contiguous views, partial final groups) must end with exactly the metric table of the
    one-rank run: same controllers, same draws by element, per-controller reductions in a fixed order => same bits."""
    args = ["--config", str(config), "--steps", str(steps), "--warmup", "3", "--no-end-to-end", "--no-cpu-baseline", "--no-also"]
    one = _self_launched(args, 1)
    three = _self_launched(args, 3)
    assert one["n_gpus"] == 1 and three["n_gpus"] == 3 and three["scaling"] == "strong"
    assert three["config"]["rccl"]["world"] == 3 and three["check"]["gather_ok"]
    for d in (one, three):
        assert d["check"]["metric_table_shape"] == [15, total] and d["check"]["metric_table_finite"]
        assert d["check"]["max_abs_err_vs_oracle"] < 1e-10 and d["check"]["rim_err"] < 1e-10
    assert three["check"]["metric_table_sha256"] == one["check"]["metric_table_sha256"]
    assert three["config"]["evals_per_step"] == one["config"]["evals_per_step"]
    assert three["roofline"]["evals_per_launch"] == (total - 2 * (total // 3)) * (10**5 if config in (4, 40) else 10**4)   # rank 0's shard


def test_bench_self_launch_many_ranks():
    """`python3 bench.py --gpus 4 --steps 20 --warmup 5`: FOUR self-launched ranks sharing the one GPU (gloo rehearsal; the
    pool's process guard allows six GPU processes at once - this pytest process is one of them - so the eight-rank case the
    driver runs on an eight-GPU node cannot be started here).  One invocation yields the weak-scaling headline, the metric's
    own workload under strong scaling, and both exchange variants of config 4; the wall time is printed for DESIGN.md 5."""
    import time
    t0 = time.time()
    d = _self_launched(["--steps", "20", "--warmup", "5"], 4)
    wall = time.time() - t0
    print(f"bench.py --gpus 4 (gloo, one GPU): wall {wall:.0f} s")
    rc = d["config"]["rccl"]
    assert d["n_gpus"] == 4 and rc["world"] == 4 and sorted(x["rank"] for x in rc["devices"]) == [0, 1, 2, 3]
    assert d["scaling"] == "weak" and d["config"]["evals_per_step"] == 4 * 10**6 and d["extras_failed"] == []
    assert d["check"]["gather_ok"] and d["check"]["metric_table_shape"] == [15, 400]
    c3 = d["also"]["config3_strong"]
    assert c3["scaling"] == "strong" and c3["n_gpus"] == 4 and c3["evals_per_step"] == 10**6 and c3["evals_per_launch"] == 25 * 10**4
    assert c3["check"]["gather_ok"] and c3["check"]["max_abs_err_vs_oracle"] < 1e-10 and c3["check"]["metric_table_shape"] == [15, 100]
    for legname in ("config4_strong", "config4_strong_fused", "config4_strong_gather_fid"):
        c4 = d["also"][legname]
        assert c4["n_gpus"] == 4 and c4["check"]["gather_ok"] and c4["evals_per_launch"] == 25 * 10**6, legname
    assert wall < 600          # the driver's limit for one bench invocation

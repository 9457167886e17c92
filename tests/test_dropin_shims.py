"""The import shims under dropin/ export exactly what INTEGRATION.md section B says - no more, no less - and the names
the reference's figure scripts import that are NOT provided are documented there (round 1's docs claimed those scripts
ran unchanged through the shim; they do not: they run per INTEGRATION.md section A, on caches produced here)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what each shim must export (INTEGRATION.md, table of section B)
SHIM_EXPORTS = {
    "mcsim": ["MCDataSim", "ExperimentNamer", "DirectoryDoesNotExistError", "wd_from_ideal", "compute_dkw_error",
              "Q", "wc_fids", "std_fids", "Q_fids", "wd_from_ideal_fids", "Q_partial", "get_cdf"],      # (+ __metric_name_to_metric__: dunder, checked below)
    "noise_model": ["noise_function", "noise_model_base", "structured_perturbation", "directional_perturbation"],
    "wd_sortof_fast_implementation": ["wd_from_ideal", "wd_from_ideal_zero", "RIM_p", "compute_dkw_error", "dkw_ecdf_bounds"],
    "noise_analysis": ["ExperimentNamer", "DirectoryDoesNotExistError"],
}

# `from <module> import <names>` lines of the reference's scripts that touch the four shimmed modules (data, with the
# reference location of each line); True = resolvable through dropin/
REFERENCE_IMPORT_LINES = [
    ("gen_fig_8_arim_fcall_scaling.py:9", "mcsim", ["MCDataSim"], True),
    ("generate_arim_all_fig5.py:9", "mcsim", ["MCDataSim"], True),
    ("generate_arim_all_fig5.py:10", "wd_sortof_fast_implementation", ["wd_from_ideal_zero"], True),
    ("generate_example_fig1.py:2", "wd_sortof_fast_implementation", ["wd_from_ideal", "dkw_ecdf_bounds"], True),
    ("exploring_rimk.py:1", "mcsim", ["MCDataSim"], True),
    ("exploring_rimk.py:7", "mcsim", ["remove_redundant_ticks"], False),
    ("generate_fig3.py:1", "mcsim", ["MCDataSim", "remove_redundant_ticks", "vn_test"], False),
    ("generate_fig3.py:6", "wd_sortof_fast_implementation", ["wd_from_ideal_zero"], True),
    ("generate_fig4_kendallrankanalysis.py:1", "mcsim", ["MCDataSim", "remove_redundant_ticks", "vn_test"], False),
    ("mcsim.py:24", "wd_sortof_fast_implementation", ["wd_from_ideal", "compute_dkw_error"], True),
    ("mcsim.py:25", "noise_model", ["structured_perturbation"], True),
    ("mcsim.py:26", "noise_analysis", ["ExperimentNamer", "DirectoryDoesNotExistError"], True),
    ("qnewton.py:18", "wd_sortof_fast_implementation", ["wd_from_ideal"], True),
]
NOT_PROVIDED_NAMES = ["remove_redundant_ticks", "vn_test"]
NOT_PROVIDED_METHODS = ["get_top_k_by_fid", "get_top_k_by_fid_idx", "get_best_controller_perf", "save_fig", "get_wd_data_c",
                        "bootstrap_resampling_std"]
PROVIDED_METHODS = ["get_fid_dists", "get_algo_fid_dist", "get_metrics_dict", "get_mcname", "load_controllers", "loadsimdata",
                    "get_controller_fid_dist_boot", "get_ranks", "set_fig_save_directory", "get_path", "merge_mcdata",
                    "merge_controller_files", "get_rims", "get_arims"]


def _exports(module: str):
    code = ("import sys, json; sys.path.insert(0, %r); import %s as m; "
            "print(json.dumps(sorted(n for n in dir(m) if not n.startswith('_'))))" % (os.path.join(ROOT, "dropin"), module))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_shims_export_exactly_the_documented_names():
    for module, names in SHIM_EXPORTS.items():
        assert _exports(module) == sorted(names), module


def test_reference_import_lines_resolve_as_documented():
    for where, module, names, ok in REFERENCE_IMPORT_LINES:
        missing = [n for n in names if n not in SHIM_EXPORTS[module]]
        assert (not missing) == ok, (where, missing)
        assert all(n in NOT_PROVIDED_NAMES for n in missing), (where, missing)


def test_mcdatasim_method_surface_and_docs():
    import importlib
    mc = importlib.import_module("code-robchar_amd.mc_data_sim").MCDataSim
    for name in PROVIDED_METHODS:
        assert callable(getattr(mc, name)), name
    for name in NOT_PROVIDED_METHODS:
        assert not hasattr(mc, name), f"{name} exists now: move it to PROVIDED_METHODS and update INTEGRATION.md"
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in NOT_PROVIDED_NAMES + NOT_PROVIDED_METHODS + PROVIDED_METHODS:
        assert name in doc, f"INTEGRATION.md does not mention {name}"
    readme = open(os.path.join(ROOT, "README.md")).read()
    assert "PYTHONPATH=dropin python generate_fig3.py" not in readme


def test_mcsim_shim_exports_the_metric_table():
    import json
    code = ("import sys, json; sys.path.insert(0, %r); import mcsim as m; "
            "print(json.dumps(list(m.__metric_name_to_metric__)))" % os.path.join(ROOT, "dropin"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    names = json.loads(out.stdout.strip().splitlines()[-1])
    assert names == [r'$W(.,\delta(x-1))$', "Q th. 0.95", "Q th. 0.98", "std", "worst case fid"]      # mcsim.py:178-183

"""Build-time resource check of the HIP kernels (CPU test: hipcc cross-compiles gfx950 without a GPU).

`make -C code-robchar_amd/csrc asm` emits the device ISA with the per-kernel metadata (.vgpr_count,
.vgpr_spill_count, .private_segment_fixed_size).  A spilled VGPR in a fidelity kernel is scratch traffic in the
innermost loop of the hot path (round 1 shipped ends-mode instantiations for N = 8, 10, 11, 12, 15, 16 that spilled,
because the residency table asked for more waves than their registers allowed) - so no instantiation of
`mc_fid_chain_kernel`, no reduction / sort / draw kernel may spill or use scratch memory.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "code-robchar_amd", "csrc")


def _kernel_resources():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    subprocess.run(["make", "-j4", "-C", CSRC, "asm"], check=True, capture_output=True)
    text = open(os.path.join(CSRC, "robchar_hip.gfx950.s")).read()
    meta = text[text.index("amdhsa.kernels:"):]
    out = {}
    for block in meta.split("  - .agpr_count:")[1:]:
        get = lambda key: re.search(r"\.%s:\s+(\S+)" % key, block).group(1)
        name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip() or get("name")
        out[name] = {k: int(get(k)) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                               "private_segment_fixed_size")}
    return out


@pytest.fixture(scope="module")
def resources():
    return _kernel_resources()


def test_every_chain_instantiation_is_present_and_spill_free(resources):
    chain = {k: v for k, v in resources.items() if "mc_fid_chain_kernel<" in k}
    # N = 2..16 x {rows, adjugate, ends} less the two end-to-end instantiations that are never dispatched (N = 15, 16: the general
    # adjugate one is faster there), plus the general adjugate instantiation for N = 17..24 (round 5)
    assert len(chain) == 15 * 3 - 2 + 8, sorted(chain)
    size = lambda k: int(re.search(r"mc_fid_chain_kernel<(\d+),", k).group(1))
    bad = {k: v for k, v in chain.items() if size(k) <= 16 and (v["vgpr_spill_count"] or v["private_segment_fixed_size"])}
    assert not bad, bad
    # N = 17..24 run one wave per SIMD on all 512 registers (256 of them accumulation registers used as a second file); what
    # does not fit even there - 6 N doubles of weight-recurrence state + the unrolled QL - spills a few dozen dwords to
    # scratch: bounded here, and measured (profiles/r05_long_chain_sweep.txt: still 5-6x faster than the LDS kernel they replace)
    big = {k: v for k, v in chain.items() if size(k) >= 17}
    assert len(big) == 8 and all(v["vgpr_spill_count"] <= 96 and v["private_segment_fixed_size"] <= 512 for v in big.values()), big


def test_philox_fused_instantiations_are_present_and_spill_free(resources):
    """mc_fid_chain_philox_kernel (round 4: the draws generated inside the fidelity kernel): N = 2..16 x {adjugate, ends}; the
    pair values must live in registers (a select between array elements once put them in scratch)."""
    fused = {k: v for k, v in resources.items() if "mc_fid_chain_philox_kernel<" in k}
    assert len(fused) == 15 * 2, sorted(fused)
    bad = {k: v for k, v in fused.items() if v["vgpr_spill_count"] or v["private_segment_fixed_size"]}
    assert not bad, bad


def test_no_vgpr_spills_anywhere(resources):
    """(except the one-wave chain instantiations for N = 17..24: see above)"""
    big = re.compile(r"mc_fid_chain_kernel<(17|18|19|20|21|22|23|24), 1>")
    bad = {k: v for k, v in resources.items() if v["vgpr_spill_count"] and not big.search(k)}
    assert not bad, bad


def test_streaming_kernels_use_no_scratch(resources):
    for frag in ("reduce_kernel", "reduce_rows_wave_kernel", "sort_", "philox_normal_kernel", "rim_p_kernel",
                 "mc_fid_chain_anyn_kernel", "mc_fid_jacobi_kernel", "mc_fid_ring_kernel", "mc_fid_ring_mixed_kernel",
                 "mc_fid_ring_repair_kernel", "mc_fid_csym_kernel", "dir_len_kernel", "dir_emit_kernel", "mt19937", "legacy_"):
        for name, res in resources.items():
            if frag in name:
                assert res["private_segment_fixed_size"] == 0, (name, res)


def test_headline_kernel_register_budget(resources):
    """BASELINE c3's kernel (N = 7, ends mode) must keep 4 waves per SIMD (<= 128 VGPRs) - measured insensitive to
    3 / 4 / 5 waves (56.5 / 57.0 / 57.5 us; the 5-wave build of the mixed-precision path spills 6 VGPRs) - and the
    mixed-precision path must really be in it: fp32 rotations (v_rsq_f32) next to the fp64 Halley step."""
    (name, res), = [(k, v) for k, v in resources.items() if "mc_fid_chain_kernel<7, 2>" in k]
    assert res["vgpr_count"] <= 128, res
    text = open(os.path.join(CSRC, "robchar_hip.gfx950.s")).read()
    start = text.index("mc_fid_chain_kernelILi7ELi2E")
    body = text[text.index(":", start):text.index(".amdhsa_kernel", start)]
    assert body.count("v_rsq_f32") >= 20 and body.count("v_fma_f64") + body.count("v_fmac_f64") >= 200


def test_legacy_stream_attempt_is_not_fma_contracted():
    """The accept / reject decision of the polar method must be bit-identical to NumPy's C code, which has no fused
    multiply-add: r2 = x1*x1 + x2*x2 has to compile to two multiplications and an addition (hipcc contracts by default).
    In the counting kernel every v_fma_f64 must be one of the exact forms (a * 2^26 + b, 2u - 1: constant operands)."""
    _kernel_resources()                                   # builds the ISA
    text = open(os.path.join(CSRC, "robchar_hip.gfx950.s")).read()
    start = text.index("legacy_count_kernel")
    body = text[text.index(":", start):text.index("s_endpgm", start)]
    fmas = [l for l in body.splitlines() if "v_fma_f64" in l]
    assert fmas and all(re.search(r"(, 2\.0, -1\.0|0x[0-9a-f]+|s\[\d+:\d+\])", l) for l in fmas), fmas
    assert body.count("v_mul_f64") >= 2 and body.count("v_add_f64") >= 1

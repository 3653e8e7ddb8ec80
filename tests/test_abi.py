"""CPU-side checks of the drop-in boundary: the shared library builds, loads and exports
every symbol include/ganspace_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ganspace_amd import _build, _lib
    _build.build()
    return _lib.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ganspace_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(lib):
    from ganspace_amd import _lib
    declared = _declared_symbols()
    assert declared, "no symbols parsed from the header"
    assert sorted(_lib.SIGNATURES) == declared
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"


def test_version_and_error_paths_without_gpu(lib):
    import ctypes as C
    assert lib.gs_version() == 1
    h = C.c_void_p()
    # argument validation happens before any HIP call
    assert lib.gs_ipca_create(0, 1, 0, 0, 0, C.byref(h)) == -1
    assert b"feature dim" in lib.gs_last_error()
    assert lib.gs_ipca_create(16, 32, 0, 0, 0, C.byref(h)) == -1       # k > d
    assert lib.gs_ipca_create(16, 4, 7, 0, 0, C.byref(h)) == -1        # bad mode
    assert lib.gs_ipca_create(16, 4, 0, 9, 0, C.byref(h)) == -5        # precision not implemented
    assert lib.gs_ipca_create(100000, 4, 0, 0, 0, C.byref(h)) == -5    # d beyond the Gram-side solver
    assert lib.gs_ipca_update(None, None, 1, 1, None) == -1
    assert lib.gs_ipca_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ganspace_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("GANSPACE_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_estimator_factory_names():
    from ganspace_amd.estimators import get_estimator
    assert get_estimator("ipca", 80, 1.0).get_param_str() == "ipca_c80"
    assert get_estimator("ipca-exact", 20, 1.0).get_param_str() == "ipca-exact_c20"
    assert get_estimator("ipca", 80, 1.0).batch_support is True
    with pytest.raises(RuntimeError, match="Unknown estimator"):
        get_estimator("bogus", 3, 1.0)
    # every name the reference's factory accepts (estimators.py:206-218) resolves, with the reference's cache keys
    # (estimators.py:28-29, 62-63, 91-92, 132-133, 178-179)
    assert get_estimator("pca", 3, 1.0).get_param_str() == "pca-full_c3"
    assert get_estimator("fbpca", 3, 1.0).get_param_str() == "fbpca_c3_it2_l6"
    assert get_estimator("ica", 3, 1.0).get_param_str() == "ica_c3_w"
    assert get_estimator("spca", 3, 2.5).get_param_str() == "spca_c3_a2.5"
    for name in ("pca", "fbpca", "ica", "spca"):
        assert get_estimator(name, 3, 1.0).batch_support is False


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ganspace_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                # no CPU linear algebra in the product (scipy.stats.truncnorm for BigGAN z parity is fine).  The one
                # exception is the separate pass-through module for the reference's non-PCA estimators ('ica',
                # 'spca': scikit-learn fits that SURVEY.md 2 keeps on the CPU), which only get_estimator() reaches.
                if f == "cpu_estimators.py":
                    continue
                assert not re.search(r"^\s*(from|import)\s+(sklearn|scipy\.linalg|scipy\.sparse)\b", txt,
                                     flags=re.M), f
                if f != "estimators.py":
                    assert "cpu_estimators" not in txt, f


def test_hot_kernels_have_no_scratch_and_products_use_the_f64_matrix_pipe(lib):
    """The build records every kernel's resources (``-Rpass-analysis=kernel-resource-usage`` ->
    ``ganspace_amd/lib/kernel_usage.json``) and FAILS when a kernel on the no-scratch list spills: the LDS-DMA Gram kernel
    counts its VMEM operations (``s_waitcnt vmcnt(N)``), so a scratch reload makes it wrong, not slow (round-3 finding).
    Here: the record exists, covers the hot kernels, and none of them touches scratch."""
    import json
    from ganspace_amd import _build
    path = os.path.join(os.path.dirname(_build.lib_path()), "kernel_usage.json")
    if not os.path.exists(path):
        _build.build(force=True)
    usage = json.load(open(path))
    hot = {k: u for k, u in usage.items() if _build.NO_SCRATCH.search(k)}
    for needle in ("gram_f32_wide_kernel", "gram_bf16_glds_kernel", "gram_partial_kernel", "rowgram_dma_kernel", "tn_gemm_kernel", "project_rows_kernel",
                   "linear_act_fast_kernel", "mm64_kernel", "chol_inv_kernel"):
        assert any(needle in k for k in hot), f"{needle} missing from the resource record"
    for k, u in hot.items():
        assert u.get("scratch", 0) == 0 and u.get("vgpr_spill", 0) == 0, (k, u)
    # the guard itself: a spilling kernel on the list must fail the build
    remarks = ("x.hip:1:1: remark: Function Name: _ZN2gs11mm64_kernelILb1ELi32EEEv [-Rpass-analysis=kernel-resource-usage]\n"
               "x.hip:1:1: remark:     VGPRs: 128 [-Rpass-analysis=kernel-resource-usage]\n"
               "x.hip:1:1: remark:     ScratchSize [bytes/lane]: 24 [-Rpass-analysis=kernel-resource-usage]\n"
               "x.hip:1:1: remark:     VGPRs Spill: 6 [-Rpass-analysis=kernel-resource-usage]\n")
    parsed = _build._kernel_usage(remarks)
    (name, u), = parsed.items()
    assert _build.NO_SCRATCH.search(name) and u["scratch"] == 24 and u["vgpr_spill"] == 6


def test_measurement_switches_are_compiled_out_of_the_production_library():
    """The A/B switches of the kernels (``gs_knob`` in csrc/gs_common.h) read the environment only in the side-by-side
    measurement build (``-DGS_MEASURE_BUILD`` -> ``ganspace_amd/lib_measure/``): no ``getenv`` of a ``GS_*`` switch is left
    in the sources outside that macro."""
    csrc = os.path.join(ROOT, "ganspace_amd", "csrc")
    offenders = []
    for f in sorted(os.listdir(csrc)):
        txt = open(os.path.join(csrc, f)).read()
        for m in re.finditer(r"\bgetenv\(\"(GS_[A-Z0-9_]+)\"\)", txt):
            # gs_common.h defines gs_knob itself; GS_GRAM_ABLATE sits behind its own -DGS_GRAM_ABLATE_BUILD
            if f == "gs_common.h" or m.group(1) == "GS_GRAM_ABLATE":
                continue
            offenders.append((f, m.group(1)))
    assert not offenders, offenders
    common = open(os.path.join(csrc, "gs_common.h")).read()
    assert "#ifdef GS_MEASURE_BUILD" in common and "inline const char *gs_knob(const char *) { return nullptr; }" in common

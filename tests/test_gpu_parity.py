"""GPU parity tests (run on a real MI355X with ``-m gpu``): the HIP path, called through
the C-ABI, against the oracle and the reference-generated golden fixtures."""
import os

import numpy as np
import pytest

import inputs as gin
from oracle import ipca as O
from oracle import synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; the product path has no CPU fallback")
    from ganspace_amd import _lib
    _lib.load()          # fail loudly if the extension is missing
    return torch.device("cuda", 0)


def _golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"ipca_ref_{name}.npz"), allow_pickle=False)


# ---- Gram / column-sum kernel --------------------------------------------------------------

@pytest.mark.parametrize("rows,d,ld_extra,use_shift", [
    (1000, 64, 0, False), (777, 200, 8, True), (2000, 512, 0, True), (333, 37, 3, True),
    (10000, 512, 0, False), (4097, 384, 0, True), (5, 128, 0, False), (30000, 96, 0, True),
    # multi-block launches: chunks longer than one float32 accumulation span (1024 rows) carry into float64 registers
    (70000, 512, 0, True), (200000, 64, 0, False), (50001, 200, 8, True),
])
def test_gram_accumulate_matches_float64(dev, rows, d, ld_extra, use_shift):
    from ganspace_amd import ops
    rs = np.random.RandomState(rows + d)
    Xh = (rs.standard_normal((rows, d + ld_extra)) * rs.uniform(0.2, 3.0, d + ld_extra) + 0.7).astype(np.float32)
    X = torch.from_numpy(Xh).to(dev)[:, :d]
    sh = (Xh[:, :d].mean(0) + 0.01).astype(np.float32) if use_shift else None
    G, cs = ops.gram_accumulate(X, shift=torch.from_numpy(sh).to(dev) if use_shift else None)
    Xc = Xh[:, :d].astype(np.float64) - (sh.astype(np.float64) if use_shift else 0.0)
    Gref = Xc.T @ Xc
    scale = np.sqrt(np.outer(np.diag(Gref), np.diag(Gref)))
    # float32 fma chains of <= 512 rows, float64 across chunks: ~1e-6 of the Cauchy-Schwarz scale
    assert np.abs(G.cpu().numpy() - Gref).max() <= 2e-6 * scale.max()
    assert np.abs((G.cpu().numpy() - Gref) / scale).max() <= 5e-6
    np.testing.assert_allclose(cs.cpu().numpy(), Xc.sum(0), atol=2e-6 * np.abs(Xc).sum(0).max())
    Gh = G.cpu().numpy()
    np.testing.assert_array_equal(Gh, Gh.T)      # exactly symmetric by construction


@pytest.mark.parametrize("d", [384, 100])
def test_gram_row_count_sweep(dev, d):
    """Chunk / stage boundaries of the launch geometry: rows are dealt to chunks in 16-row units, the first LDS stage
    of a chunk takes its length mod 64, the sub-16 tail of the last chunk is masked - every combination must give
    the same sums (this sweep caught an odd-length first stage double counting a row)."""
    from ganspace_amd import ops
    rs = np.random.RandomState(d)
    for rows in (1, 2, 15, 16, 17, 31, 33, 48, 49, 63, 64, 65, 81, 96, 111, 113, 177, 1023, 1600, 4095, 4097, 6401):
        Xh = rs.standard_normal((rows, d)).astype(np.float32)
        X = torch.from_numpy(Xh).to(dev)
        Gref = Xh.astype(np.float64).T @ Xh.astype(np.float64)
        for precision in ("f32", "bf16x6"):
            G, cs = ops.gram_accumulate(X, precision=precision)
            err = np.abs(G.cpu().numpy() - Gref).max() / max(np.abs(Gref).max(), 1e-30)
            assert err <= 2e-6, (rows, d, precision, err)
            np.testing.assert_allclose(cs.cpu().numpy(), Xh.astype(np.float64).sum(0), atol=2e-5 * max(1.0, rows ** 0.5))


def test_gram_block_is_a_narrow_slice_of_very_wide_rows(dev):
    """ld >> d with more than 4 GB between the first and the last row: the buffer-resource fast path (32-bit byte
    offsets) must hand over to the general path."""
    from ganspace_amd import ops
    rows, d, ld = 18000, 256, 60000                       # 4.3 GB of float32
    g = torch.Generator(device=dev).manual_seed(3)
    big = torch.empty((rows, ld), dtype=torch.float32, device=dev)
    big[:, :d] = torch.randn((rows, d), generator=g, device=dev)
    X = big[:, :d]
    G, cs = ops.gram_accumulate(X)
    Xd = X.double()
    Gref = (Xd.T @ Xd)
    assert float((G - Gref).abs().max() / Gref.abs().max()) <= 2e-6
    assert float((cs - Xd.sum(0)).abs().max()) <= 1e-2
    del big


def test_gram_accumulates_and_is_linear(dev):
    from ganspace_amd import ops
    rs = np.random.RandomState(5)
    X = torch.from_numpy(rs.standard_normal((3000, 256)).astype(np.float32)).to(dev)
    G1, c1 = ops.gram_accumulate(X)
    G2, c2 = ops.gram_accumulate(X, G1.clone(), c1.clone())
    torch.testing.assert_close(G2, 2 * G1, rtol=1e-12, atol=0)
    torch.testing.assert_close(c2, 2 * c1, rtol=1e-12, atol=0)
    # splitting the rows changes only float32 chunk boundaries
    Ga, ca = ops.gram_accumulate(X[:1234])
    Gb, cb = ops.gram_accumulate(X[1234:], Ga, ca)
    assert (Gb - G1).abs().max().item() <= 2e-6 * G1.abs().max().item()


# ---- split-bf16 contraction modes (opt-in precision) ---------------------------------------------

# (d = 512 launches of >= 20 000 rows take the "wide" bf16x3 kernel - pairs of workgroups holding the whole upper
#  triangle: a multiple of the k-step, a ragged end, more than one launch, an aligned padded stride and an unaligned
#  one that has to stay with the tiled kernel)
@pytest.mark.parametrize("rows,d,ld_extra", [(1, 4, 0), (17, 100, 3), (1000, 512, 0), (10000, 512, 0),
                                             (2311, 640, 0), (4097, 96, 5), (20000, 512, 0), (50001, 512, 4),
                                             (131072 + 20777, 512, 0), (30003, 512, 3)])
@pytest.mark.parametrize("precision,tol", [("bf16x6", 3e-6), ("bf16x3", 6e-5), ("bf16", 6e-3)])
def test_gram_split_bf16_matches_float64(dev, rows, d, ld_extra, precision, tol):
    """x = hi + mid (+ lo) in bf16, products rebuilt from 3 / 6 bf16 MFMAs: dropped terms are 2^-16 / 2^-24 of
    |x||y| per product, so the elementwise error is bounded on the Cauchy-Schwarz scale.  Plain ``bf16`` (one term,
    one MFMA): each operand is off by <= 2^-9 relative, a product by <= 2^-8 (4e-3), whatever the row count."""
    from ganspace_amd import ops
    rs = np.random.RandomState(rows + d)
    Xh = (rs.standard_normal((rows, d + ld_extra)) * rs.uniform(0.2, 3.0, d + ld_extra) + 0.7).astype(np.float32)
    X = torch.from_numpy(Xh).to(dev)[:, :d]
    sh = (Xh[:, :d].mean(0) + 0.01).astype(np.float32)
    G, cs = ops.gram_accumulate(X, shift=torch.from_numpy(sh).to(dev), precision=precision)
    Xc = Xh[:, :d].astype(np.float64) - sh.astype(np.float64)
    Gref = Xc.T @ Xc
    scale = np.sqrt(np.outer(np.diag(Gref), np.diag(Gref)))
    Gh = G.cpu().numpy()
    assert np.abs((Gh - Gref) / scale).max() <= tol, np.abs((Gh - Gref) / scale).max()
    np.testing.assert_allclose(cs.cpu().numpy(), Xc.sum(0), atol=2e-6 * np.abs(Xc).sum(0).max())
    np.testing.assert_array_equal(Gh, Gh.T)


def test_gram_split_bf16x6_is_float32_class(dev):
    """The six-product mode is as close to the float64 Gram as the exact-f32 MFMA path (same slabs, same f64 fold)."""
    from ganspace_amd import ops
    rs = np.random.RandomState(11)
    Xh = (rs.standard_normal((10000, 512)) * rs.uniform(0.1, 4.0, 512)).astype(np.float32)
    X = torch.from_numpy(Xh).to(dev)
    Gref = Xh.astype(np.float64).T @ Xh.astype(np.float64)
    scale = np.sqrt(np.outer(np.diag(Gref), np.diag(Gref)))
    e32 = np.abs((ops.gram_accumulate(X)[0].cpu().numpy() - Gref) / scale).max()
    e6 = np.abs((ops.gram_accumulate(X, precision="bf16x6")[0].cpu().numpy() - Gref) / scale).max()
    e3 = np.abs((ops.gram_accumulate(X, precision="bf16x3")[0].cpu().numpy() - Gref) / scale).max()
    assert e6 <= max(2.0 * e32, 1e-6), (e32, e6)
    assert e3 <= 6e-5, e3


@pytest.mark.parametrize("name", ["d64_k8", "d512_k20", "d512_k80_nb10000", "d96_k12_bigmean"])
@pytest.mark.parametrize("precision", ["bf16x6", "bf16x3"])
def test_exact_mode_split_bf16_matches_exact_oracle(dev, golden_dir, name, precision):
    from ganspace_amd.estimators import IPCAEstimator
    case = gin.IPCA_CASES[name]
    est = IPCAEstimator(case["k"], "exact", precision=precision)
    for X in gin.ipca_blocks(case):
        assert est.fit_partial(torch.from_numpy(X).to(dev)) is True
    ex = O.exact_pca(gin.ipca_blocks(case), case["k"])
    t = est.transformer
    r = case["ncheck"]
    cos = O.signed_cosines(t.components_[:r], ex["components_"][:r])
    x6 = precision == "bf16x6"
    assert cos.min() > 1 - (1e-6 if x6 else 2e-5), cos.min()
    np.testing.assert_allclose(t.singular_values_[:r], ex["singular_values_"][:r], rtol=2e-5 if x6 else 2e-4)
    np.testing.assert_allclose(t.mean_, ex["mean_"], atol=2e-6 * max(1.0, np.abs(ex["mean_"]).max()))
    g = _golden(golden_dir, name)
    top = min(20, max(1, case["k"] // 4))
    assert np.abs(O.signed_cosines(t.components_[:top], g["components"][:top])).min() > 0.999


@pytest.mark.parametrize("name", ["d512_k20", "d200_k200_ragged"])
def test_faithful_mode_split_bf16x6_matches_reference_fixture(dev, golden_dir, name):
    from ganspace_amd.estimators import IPCAEstimator
    case = gin.IPCA_CASES[name]
    g = _golden(golden_dir, name)
    est = IPCAEstimator(case["k"], "faithful", precision="bf16x6")
    for X in gin.ipca_blocks(case):
        assert est.fit_partial(torch.from_numpy(X).to(dev)) is True
    r = case["ncheck"]
    cos = O.signed_cosines(est.transformer.components_[:r], g["components"][:r])
    assert cos.min() > 1 - 5e-6, cos.min()
    np.testing.assert_allclose(est.transformer.singular_values_[:r], g["singular_values"][:r], rtol=1e-4)


@pytest.mark.parametrize("precision,cos_tol,sv_tol", [("bf16x6", 5e-6, 1e-4), ("bf16x3", 2e-4, 2e-3)])
@pytest.mark.parametrize("name", ["d3000_k12_highd", "d512_k20"])
def test_smallside_split_bf16_matches_reference_fixture(dev, golden_dir, name, precision, cos_tol, sv_tol):
    """Small-side recurrence with T = M M^T on the bf16 matrix cores (both operands are K-contiguous rows of M).
    bf16x6 drops terms below 2^-24 |xy| per product: float32-class, same tolerances as the f32 contraction;
    bf16x3 drops terms below 2^-16 |xy| (zero-mean rounding errors, averaged over d columns)."""
    from ganspace_amd.estimators import IPCAEstimator
    case = gin.IPCA_CASES[name]
    g = _golden(golden_dir, name)
    est = IPCAEstimator(case["k"], "smallside", precision=precision)
    for X in gin.ipca_blocks(case):
        assert est.fit_partial(torch.from_numpy(X).to(dev)) is True
    comp, stdev, ratio = est.get_components()
    r = case["ncheck"]
    cos = O.signed_cosines(comp[:r], g["components"][:r])
    assert cos.min() > 1 - cos_tol, (name, precision, cos.min())
    np.testing.assert_allclose(est.transformer.singular_values_[:r], g["singular_values"][:r], rtol=sv_tol)
    np.testing.assert_allclose(est.transformer.mean_, g["mean"], atol=2e-6 * max(1.0, np.abs(g["mean"]).max()))


# ---- eigensolver -----------------------------------------------------------------------------

@pytest.mark.parametrize("n,rank", [(8, 8), (64, 64), (129, 129), (200, 50), (512, 512)])
def test_eigh_matches_lapack(dev, n, rank):
    from ganspace_amd import ops
    rs = np.random.RandomState(n)
    B = rs.standard_normal((rank, n)) * (1.15 ** -np.arange(rank))[:, None]
    A = B.T @ B
    w, V, sweeps = ops.eigh_sym(torch.from_numpy(A).to(dev))
    w, V = w.cpu().numpy(), V.cpu().numpy()
    wr, Ur = np.linalg.eigh(A)
    wr, Ur = wr[::-1], Ur[:, ::-1].T
    assert sweeps < 40
    np.testing.assert_allclose(w, np.maximum(wr, 0), atol=1e-12 * wr[0])
    r = min(rank, 40)
    cos = np.abs(np.sum(V[:r] * Ur[:r], axis=1))
    assert cos.min() > 1 - 1e-10
    # orthonormal rows on the numerically non-null part, residual A v = w v
    Vr = V[:r]
    assert np.abs(Vr @ Vr.T - np.eye(r)).max() < 1e-11
    assert np.abs(A @ Vr.T - Vr.T * w[:r]).max() < 1e-11 * wr[0]


# ---- incremental PCA, sklearn-faithful mode vs the reference fixtures ----------------------------

def _fit_device(name_or_case, mode, dev, device_input=True):
    from ganspace_amd.estimators import get_estimator
    case = gin.IPCA_CASES[name_or_case] if isinstance(name_or_case, str) else name_or_case
    est = get_estimator("ipca" if mode == "faithful" else "ipca-exact", case["k"], 1.0)
    for X in gin.ipca_blocks(case):
        arg = torch.from_numpy(X).to(dev) if device_input else X
        assert est.fit_partial(arg) is True
    return est


@pytest.mark.parametrize("name", list(gin.IPCA_CASES))
def test_faithful_mode_matches_reference_fixture(dev, golden_dir, name):
    case = gin.IPCA_CASES[name]
    g = _golden(golden_dir, name)
    est = _fit_device(name, "faithful", dev, device_input=(name != "d64_k8"))
    comp, stdev, ratio = est.get_components()
    t = est.transformer
    r = case["ncheck"]
    assert comp.shape == (case["k"], case["d"]) and comp.dtype == np.float32
    cos = O.signed_cosines(comp[:r], g["components"][:r])
    # tolerance: the reference's block 1 runs a float32 SVD; ours is f32-Gram + f64 eigh
    assert cos.min() > 1 - 5e-6, (name, cos.min())
    scale = g["singular_values"][0]
    np.testing.assert_allclose(t.singular_values_[:r], g["singular_values"][:r], rtol=1e-4)
    # singular values below sqrt(eps_f32) * sigma_max (numerically-null directions) are not
    # resolvable from a float32-accumulated Gram: lambda error ~1e-7 lambda_max -> sigma ~3e-4 sigma_max
    np.testing.assert_allclose(t.singular_values_, g["singular_values"], atol=5e-4 * scale)
    np.testing.assert_allclose(t.mean_, g["mean"], atol=2e-6 * max(1.0, np.abs(g["mean"]).max()))
    np.testing.assert_allclose(t.var_, g["var"], rtol=1e-4)
    np.testing.assert_allclose(stdev[:r], g["stdev"][:r], rtol=1e-4)
    np.testing.assert_allclose(ratio[:r], g["var_ratio"][:r], rtol=2e-4)
    assert int(t.n_samples_seen_) == int(g["n_samples_seen"])
    assert est.get_param_str() == str(g["param_str"])


@pytest.mark.parametrize("name", ["d64_k8", "d512_k20", "d96_k12_bigmean", "d200_k200_ragged"])
def test_faithful_mode_matches_gram_oracle_tightly(dev, name):
    case = gin.IPCA_CASES[name]
    est = _fit_device(name, "faithful", dev)
    orc = O.IPCAEstimatorOracle(case["k"], "gram")
    for X in gin.ipca_blocks(case):
        orc.fit_partial(X)
    r = case["ncheck"]
    cos = O.signed_cosines(est.transformer.components_[:r], orc.transformer.components_[:r])
    # float32 partial sums per row chunk (the chunk boundaries are a launch-geometry detail) against the oracle's
    # float64 Gram: closely spaced tail components of the k = d case move by a few 1e-6
    assert cos.min() > 1 - (3e-6 if name == "d200_k200_ragged" else 1e-6)
    np.testing.assert_allclose(est.transformer.singular_values_[:r], orc.transformer.singular_values_[:r], rtol=1e-4)
    np.testing.assert_allclose(est.transformer.mean_, orc.transformer.mean_,
                               atol=1e-6 * max(1.0, np.abs(orc.transformer.mean_).max()))


@pytest.mark.parametrize("name", ["d64_k8", "d512_k20", "d512_k80_nb10000", "d96_k12_bigmean"])
def test_exact_mode_matches_exact_oracle(dev, golden_dir, name):
    case = gin.IPCA_CASES[name]
    est = _fit_device(name, "exact", dev)
    ex = O.exact_pca(gin.ipca_blocks(case), case["k"])
    t = est.transformer
    r = case["ncheck"]
    cos = O.signed_cosines(t.components_[:r], ex["components_"][:r])
    assert cos.min() > 1 - 1e-6, cos.min()
    np.testing.assert_allclose(t.singular_values_[:r], ex["singular_values_"][:r], rtol=2e-5)
    np.testing.assert_allclose(t.explained_variance_ratio_[:r], ex["explained_variance_ratio_"][:r], rtol=1e-4)
    np.testing.assert_allclose(t.mean_, ex["mean_"], atol=2e-6 * max(1.0, np.abs(ex["mean_"]).max()))
    np.testing.assert_allclose(t.var_, ex["var_"], rtol=1e-4)
    # BASELINE target: top-20 |cos| >= 0.999 against the reference CPU IPCA.  IPCA's own
    # truncation error grows towards component k (min |cos| 0.907 at k), so the
    # claim is made for the leading quarter of the kept components (top-20 of k=80 in cfg2).
    g = _golden(golden_dir, name)
    top = min(20, max(1, case["k"] // 4))
    assert np.abs(O.signed_cosines(t.components_[:top], g["components"][:top])).min() > 0.999
    assert est.get_param_str() == f"ipca-exact_c{case['k']}"


@pytest.mark.parametrize("name", ["d3000_k12_highd", "d512_k20", "d128_k24_lowrank", "d64_k8"])
def test_smallside_mode_matches_reference_fixture(dev, golden_dir, name):
    """d >> m formulation (T = M M^T): same recurrence, same answers as the reference fixture."""
    from ganspace_amd.estimators import IPCAEstimator
    case = gin.IPCA_CASES[name]
    g = _golden(golden_dir, name)
    est = IPCAEstimator(case["k"], "smallside")
    for X in gin.ipca_blocks(case):
        assert est.fit_partial(torch.from_numpy(X).to(dev)) is True
    comp, stdev, ratio = est.get_components()
    t = est.transformer
    r = case["ncheck"]
    cos = O.signed_cosines(comp[:r], g["components"][:r])
    assert cos.min() > 1 - 5e-6, (name, cos.min())
    scale = g["singular_values"][0]
    np.testing.assert_allclose(t.singular_values_[:r], g["singular_values"][:r], rtol=1e-4)
    np.testing.assert_allclose(t.singular_values_, g["singular_values"], atol=5e-4 * scale)
    np.testing.assert_allclose(t.mean_, g["mean"], atol=2e-6 * max(1.0, np.abs(g["mean"]).max()))
    np.testing.assert_allclose(t.var_, g["var"], rtol=1e-4)
    np.testing.assert_allclose(ratio[:r], g["var_ratio"][:r], rtol=2e-4)
    assert int(t.n_samples_seen_) == int(g["n_samples_seen"])
    assert est.get_param_str() == str(g["param_str"])


def test_smallside_is_selected_automatically_for_wide_features(dev):
    from ganspace_amd import _lib
    from ganspace_amd.estimators import get_estimator
    rs = np.random.RandomState(0)
    A = rs.standard_normal((24, 16384)) * (1.2 ** -np.arange(24))[:, None]
    est = get_estimator("ipca", 6, 1.0)
    orc = O.IPCAEstimatorOracle(6, "svd")
    for _ in range(2):
        X = (rs.standard_normal((64, 24)) @ A + 0.01 * rs.standard_normal((64, 16384))).astype(np.float32)
        assert est.fit_partial(torch.from_numpy(X).to(dev))
        orc.fit_partial(X)
    assert est.transformer._mode == _lib.GS_MODE_SMALLSIDE
    cos = O.signed_cosines(est.transformer.components_, orc.transformer.components_)
    assert cos.min() > 1 - 1e-5, cos
    np.testing.assert_allclose(est.transformer.singular_values_, orc.transformer.singular_values_, rtol=1e-4)


def test_first_block_smaller_than_k_returns_false(dev, capsys):
    from ganspace_amd.estimators import get_estimator
    est = get_estimator("ipca", 20, 1.0)
    assert est.fit_partial(np.zeros((10, 32), dtype=np.float32)) is False
    assert "IPCA error" in capsys.readouterr().out


def test_unknown_estimator_raises():
    from ganspace_amd.estimators import get_estimator
    with pytest.raises(RuntimeError):
        get_estimator("nope", 3, 1.0)


def test_caller_buffer_is_left_untouched_and_reusable(dev):
    # decomposition.py:243,261,290-291: the caller overwrites X for the next block and centres
    # the last block in place afterwards -> the estimator must not alias or modify it
    case = gin.IPCA_CASES["d64_k8"]
    from ganspace_amd.estimators import get_estimator
    est = get_estimator("ipca", case["k"], 1.0)
    buf = np.ones((300, 64), dtype=np.float32)
    for X in gin.ipca_blocks(case):
        buf[:] = X
        keep = buf.copy()
        assert est.fit_partial(buf)
        np.testing.assert_array_equal(buf, keep)
        buf[:] = -1e9          # clobber immediately, like the reference loop does
    ref = _fit_device("d64_k8", "faithful", dev)
    np.testing.assert_array_equal(est.transformer.components_, ref.transformer.components_)


def test_state_merge_equals_single_fit(dev):
    from ganspace_amd.estimators import IPCAEstimator
    from ganspace_amd import distributed as D
    case = gin.IPCA_CASES["d512_k20"]
    blocks = list(gin.ipca_blocks(case))
    full = IPCAEstimator(case["k"], "exact")
    a, b = IPCAEstimator(case["k"], "exact"), IPCAEstimator(case["k"], "exact")
    for i, X in enumerate(blocks):
        Xd = torch.from_numpy(X).to(dev)
        full.fit_partial(Xd)
        (a if i < 2 else b).fit_partial(Xd)
    merged = D.merge_states([a.transformer.export_state(), b.transformer.export_state()], case["d"])
    c = IPCAEstimator(case["k"], "exact")
    c.transformer.import_state(merged, case["d"])
    cos = O.signed_cosines(c.transformer.components_, full.transformer.components_)
    assert cos.min() > 1 - 1e-9  # same data, different float32 chunking only
    np.testing.assert_allclose(c.transformer.singular_values_, full.transformer.singular_values_, rtol=2e-6)
    np.testing.assert_allclose(c.transformer.mean_, full.transformer.mean_, atol=1e-6)
    assert int(c.transformer.n_samples_seen_) == sum(case["blocks"])


def test_transform_projects_onto_components(dev):
    case = gin.IPCA_CASES["d512_k20"]
    est = _fit_device("d512_k20", "faithful", dev)
    X = next(gin.ipca_blocks(case))[:100]
    t = est.transformer
    Y = t.transform(X)
    Yref = (X.astype(np.float64) - t.mean_) @ t.components_.astype(np.float64).T
    np.testing.assert_allclose(Y, Yref, atol=2e-4 * np.abs(Yref).max())


def test_transform_bigmean_matches_sklearn_transform(dev):
    """``est.transformer.transform(X)`` on data that sits on a mean 40 x its spread: sklearn centres
    first (``X - mean_`` in float64), and so does the device path (rows centred while the projection kernel stages them);
    ``X C^T - mean C^T`` in float32 would lose ~ |mean| / sigma x eps32 of every coordinate.  Checked against scikit-learn's
    own ``IncrementalPCA.transform`` fitted on the same blocks."""
    from sklearn.decomposition import IncrementalPCA
    case = gin.IPCA_CASES["d96_k12_bigmean"]
    est = _fit_device("d96_k12_bigmean", "faithful", dev)
    sk = IncrementalPCA(n_components=case["k"], batch_size=max(100, 2 * case["k"]))
    blocks = list(gin.ipca_blocks(case))
    for X in blocks:
        sk.partial_fit(X)
    X = blocks[-1][:257]                                    # ragged row count
    Y = est.transformer.transform(X)
    Ysk = sk.transform(X)
    assert Y.shape == Ysk.shape == (257, case["k"])
    # float32 rows of magnitude ~ |mean| = 40 sigma carry 40 eps32 sigma of representation error before anything is computed
    np.testing.assert_allclose(Y, Ysk, atol=3e-5 * np.abs(Ysk).max())
    # device tensor in, strided rows (a column slice of a wider buffer)
    wide = torch.zeros((257, 128), dtype=torch.float32, device=dev)
    wide[:, :96] = torch.from_numpy(X).to(dev)
    np.testing.assert_allclose(est.transformer.transform(wide[:, :96]), Y, atol=1e-6 * np.abs(Ysk).max())


def test_transform_accepts_any_feature_width(dev):
    """n_features not a multiple of 4 (sklearn's transform accepts any width; round 5 raised NotImplementedError)."""
    from ganspace_amd.estimators import IPCAEstimator
    rs = np.random.RandomState(5)
    d, k = 37, 6
    A = rs.standard_normal((10, d)) * (1.5 ** -np.arange(10))[:, None]
    mu = rs.standard_normal(d) * 5.0
    est = IPCAEstimator(k, "faithful")
    blocks = [(rs.standard_normal((200, 10)) @ A + mu + 0.01 * rs.standard_normal((200, d))).astype(np.float32)
              for _ in range(3)]
    for X in blocks:
        assert est.fit_partial(X)
    t = est.transformer
    X = blocks[1][:33]
    Y = t.transform(X)
    Yref = (X.astype(np.float64) - t.mean_) @ t.components_.astype(np.float64).T
    assert Y.shape == (33, k)
    np.testing.assert_allclose(Y, Yref, atol=2e-5 * np.abs(Yref).max())
    with pytest.raises(ValueError):
        t.transform(np.zeros((4, d + 1), np.float32))


# ---- z -> activation layers --------------------------------------------------------------------

def test_mapping_network_matches_oracle_and_reference_gmapping(dev, golden_dir):
    from ganspace_amd import ops
    W, b = gin.mapping_weights()
    z = gin.mapping_z()
    out = ops.mapping_forward(torch.from_numpy(z).to(dev), torch.from_numpy(W).to(dev), torch.from_numpy(b).to(dev),
                              lr_mul=gin.MAPPING_CASE["lr_mul"]).cpu().numpy()
    ref64 = synth.mapping_network(z, W, b, lr_mul=gin.MAPPING_CASE["lr_mul"])
    g = np.load(os.path.join(golden_dir, "mapping_gmapping_ref.npz"))
    scale = np.abs(ref64).max()
    # exact-f32 MFMA fma chains over K=512, 8 layers deep: float32 roundoff class
    assert np.abs(out - ref64).max() < 2e-5 * scale
    assert np.abs(out - g["w_f64"]).max() < 2e-5 * scale
    # at least as close to float64 truth as the reference's own float32 CPU run (x4 slack)
    assert np.abs(out - ref64).max() <= 4 * np.abs(g["w_f32"] - g["w_f64"]).max() + 1e-7 * scale


@pytest.mark.parametrize("rows", [30_077, 140_003])
def test_mapping_network_long_calls_equal_short_calls_bit_for_bit(dev, rows):
    """The pre-sampling phase and the Z-space fit loop push many mini-batches through ONE mapping call (80 000 rows: the
    per-layer kernel fills the chip for eight rounds instead of one partial one).  A row's result must not depend on the
    call it travels in: the tile height (32 R rows, R chosen per call length) only changes which workgroup computes an
    output element, not its k-ordered float32 fma chain - so long and short calls agree BIT FOR BIT; plus the float64 oracle
    on a sample of rows, non-zero biases, an ``out=`` row slice, and the PixelNorm-free form."""
    from ganspace_amd import ops
    W, _ = gin.mapping_weights()
    rs = np.random.RandomState(rows)
    b = (0.3 * rs.standard_normal((W.shape[0], W.shape[1]))).astype(np.float32)
    z = rs.standard_normal((rows, 512)).astype(np.float32)
    zd, Wd, bd = torch.from_numpy(z).to(dev), torch.from_numpy(W).to(dev), torch.from_numpy(b).to(dev)
    lr = gin.MAPPING_CASE["lr_mul"]
    big = torch.full((rows + 5, 512), float("nan"), device=dev)
    out = ops.mapping_forward(zd, Wd, bd, lr_mul=lr, out=big[3:3 + rows])            # a row slice of a larger array
    assert out.data_ptr() == big[3:].data_ptr()
    assert torch.isnan(big[:3]).all() and torch.isnan(big[3 + rows:]).all()         # nothing written around it
    out = out.cpu().numpy()
    assert np.isfinite(out).all()
    pick = np.r_[0:130, rows // 2:rows // 2 + 130, rows - 130:rows]
    ref64 = synth.mapping_network(z[pick], W, b, lr_mul=lr)
    assert np.abs(out[pick] - ref64).max() < 2e-5 * np.abs(ref64).max()
    short = np.concatenate([ops.mapping_forward(zd[lo:lo + 10_000], Wd, bd, lr_mul=lr).cpu().numpy()
                            for lo in range(0, rows, 10_000)])
    np.testing.assert_array_equal(out, short)
    plain = ops.mapping_forward(zd, Wd, None, lr_mul=lr, pixelnorm=False).cpu().numpy()
    refp = z[pick].astype(np.float64)
    for wl in W:                                                                     # no PixelNorm, zero biases
        refp = synth.equal_linear_lrelu(refp, wl.astype(np.float64), np.zeros(512), lr)
    assert np.abs(plain[pick] - refp).max() < 2e-5 * np.abs(refp).max()


@pytest.mark.parametrize("rows,in_f,out_f", [(64, 256, 32768), (1000, 512, 512), (130, 128, 200), (7, 64, 64)])
def test_linear_forward_matches_float64(dev, rows, in_f, out_f):
    from ganspace_amd import ops
    rs = np.random.RandomState(rows)
    x = rs.standard_normal((rows, in_f)).astype(np.float32)
    W = (rs.standard_normal((out_f, in_f)) / np.sqrt(in_f)).astype(np.float32)
    b = rs.standard_normal(out_f).astype(np.float32)
    y = ops.linear_forward(torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev), torch.from_numpy(b).to(dev))
    ref = synth.linear(x, W, b)
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("rows,d,k,use_shift", [(2000, 32768, 80, True), (333, 1028, 20, True), (1, 64, 130, False),
                                                 (129, 4100, 128, True)])
def test_project_rows_matches_float64(dev, rows, d, k, use_shift):
    """Regression projection ``((x - mean) @ comp.T) / stdev`` (reference decomposition.py:110-118) written into a
    column block of a wider staging buffer: float64 numpy is the checker; the columns next to the block stay untouched.
    The data sit on a mean 5x their spread: centring AFTER the product would lose a digit, centring first does not."""
    from ganspace_amd import ops
    rs = np.random.RandomState(rows + d)
    mean = (5.0 * rs.standard_normal(d)).astype(np.float32)
    x = (rs.standard_normal((rows, d)) + mean).astype(np.float32)
    comp = np.linalg.qr(rs.standard_normal((d, k)))[0].T.astype(np.float32) if k <= d else \
        rs.standard_normal((k, d)).astype(np.float32)
    scale = (1.0 / (0.5 + rs.rand(k))).astype(np.float32)
    buf = torch.full((rows, k + 37), -7.0, dtype=torch.float32, device=dev)
    out = ops.project_rows(torch.from_numpy(x).to(dev), torch.from_numpy(comp).to(dev),
                           shift=torch.from_numpy(mean).to(dev) if use_shift else None,
                           colscale=torch.from_numpy(scale).to(dev), out=buf[:, :k])
    assert out.data_ptr() == buf.data_ptr()
    xc = x.astype(np.float64) - (mean.astype(np.float64) if use_shift else 0.0)
    ref = (xc @ comp.astype(np.float64).T) * scale.astype(np.float64)
    got = buf.cpu().numpy()
    assert np.abs(got[:, :k] - ref).max() < 2e-5 * np.abs(ref).max()
    assert (got[:, k:] == -7.0).all()


@pytest.mark.parametrize("rows_a,rows_b,cols", [(300, 200, 96), (128, 128, 32), (1000, 512, 4608), (5, 3, 8), (257, 129, 100)])
def test_gemm_blocked_nt_matches_float64(dev, rows_a, rows_b, cols):
    """``A @ B.T`` from panel-blocked operands (csrc/gs_gemm_blocked.hip: LDS-DMA stages, the same fma chains over K as
    ``gs_linear_forward``): float64 numpy is the checker; ragged panel and K-block tails included."""
    from ganspace_amd import ops
    rs = np.random.RandomState(rows_a + cols)
    a = rs.standard_normal((rows_a, cols)).astype(np.float32)
    b = (rs.standard_normal((rows_b, cols)) / np.sqrt(cols)).astype(np.float32)
    ab, bb = ops.block_rows(torch.from_numpy(a).to(dev)), ops.block_rows(torch.from_numpy(b).to(dev))
    got = ops.gemm_blocked_nt(ab, rows_a, bb, rows_b, cols).cpu().numpy()
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("b,h,w,c,out", [(3, 5, 7, 32, 40), (2, 16, 16, 64, 128), (1, 4, 4, 512, 512)])
def test_im2col3x3_blocked_product_matches_conv2d(dev, b, h, w, c, out):
    """The blocked patch matrix of an NHWC tensor (``gs_im2col3x3_blocked``: zero padding 1, column (kh, kw, c)) times the
    blocked weight matrix IS ``F.conv2d(x, W, padding=1)`` (float64 torch on the device is the checker)."""
    import torch.nn.functional as F
    from ganspace_amd import ops
    g = torch.Generator(device="cpu").manual_seed(b * 100 + c)
    x = torch.randn(b, c, h, w, generator=g)
    wgt = torch.randn(out, c, 3, 3, generator=g) / np.sqrt(9 * c)
    xd = x.to(dev)
    cols = ops.im2col3x3_blocked(xd.permute(0, 2, 3, 1).contiguous())
    wblk = ops.block_rows(wgt.permute(0, 2, 3, 1).reshape(out, 9 * c).contiguous().to(dev))
    y = ops.gemm_blocked_nt(cols, b * h * w, wblk, out, 9 * c).view(b, h, w, out).permute(0, 3, 1, 2)
    ref = F.conv2d(xd.double(), wgt.to(dev).double(), padding=1)
    assert (y.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


def test_modulated_conv_blocked_path_equals_strided_path(dev, monkeypatch):
    """``ModulatedConv2d`` on the device: the blocked convolution path and the strided-view im2col + ``gs_linear_forward``
    path (``GANSPACE_CONV=strided``) give the same layer output to float32 roundoff, upsampling included."""
    from ganspace_amd.wrappers import ModulatedConv2d
    torch.manual_seed(11)
    m = ModulatedConv2d(64, 96, 3, 32, upsample=True).to(dev)
    x = torch.randn(6, 64, 8, 8, device=dev)
    style = torch.randn(6, 32, device=dev)
    with torch.no_grad():
        monkeypatch.setenv("GANSPACE_CONV", "blocked")
        a = m(x, style)
        monkeypatch.setenv("GANSPACE_CONV", "strided")
        b_ = m(x, style)
        ref = m.double().forward_grouped(x.double(), style.double())
    assert a.shape == b_.shape == (6, 96, 16, 16)
    assert (a - b_).abs().max().item() < 1e-5 * b_.abs().max().item()
    assert (a.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("up,h,cin,cout,b", [(False, 8, 64, 96, 5), (True, 8, 64, 96, 5), (True, 4, 512, 512, 3),
                                             (False, 4, 32, 40, 7), (True, 5, 32, 130, 2)])
def test_styled_conv_fused_launches_equal_the_unfused_layer(dev, monkeypatch, up, h, cin, cout, b):
    """``StyledConv`` on the device in two launches (``gs_modconv3x3_patches``: style modulation + bilinear x 2 upsampling
    while the patches are gathered; ``gs_gemm_blocked_nt_styled``: demodulation + noise + bias + sqrt(2) lrelu in the GEMM's
    store) against the layer as the published definition states it - per-sample weights, ``F.interpolate``, grouped
    convolution, noise injection, fused leaky ReLU - in float64, and against this repository's unfused float32 path."""
    import torch.nn.functional as F
    from ganspace_amd.wrappers import StyledConv
    torch.manual_seed(17 + h + cin)
    m = StyledConv(cin, cout, 3, 48, upsample=up).to(dev)
    with torch.no_grad():
        m.noise_weight.fill_(0.37)
        m.bias.copy_(0.2 * torch.randn_like(m.bias))
    x = torch.randn(b, cin, h, h, device=dev)
    style = torch.randn(b, 48, device=dev)
    H = 2 * h if up else h
    noise = torch.randn(1, 1, H, H, device=dev)
    with torch.no_grad():
        assert m.conv.fused_available(x)
        got = m(x, style, noise=noise)
        monkeypatch.setenv("GANSPACE_CONV_FUSED", "0")
        assert not m.conv.fused_available(x)
        unfused = m(x, style, noise=noise)
        m64 = m.double()
        ref = m64.conv.forward_grouped(x.double(), style.double())
        ref = np.sqrt(2.0) * F.leaky_relu(ref + m64.noise_weight * noise.double() + m64.bias, 0.2)
    assert got.shape == ref.shape == (b, cout, H, H)
    scale = ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() < 2e-5 * scale
    assert (got - unfused).abs().max().item() < 2e-5 * scale
    with torch.no_grad():                                            # no noise, several sub-batches (staging cap)
        from ganspace_amd import wrappers
        monkeypatch.delenv("GANSPACE_CONV_FUSED")
        monkeypatch.setattr(wrappers, "CONV_STAGING_BYTES", 2 * H * H * 9 * cin * 4)
        m32 = m.float()
        a = m32(x, style)
        ref2 = np.sqrt(2.0) * F.leaky_relu(m64.double().conv.forward_grouped(x.double(), style.double()) + m.bias.double(), 0.2)
    assert (a.double() - ref2).abs().max().item() < 2e-5 * ref2.abs().max().item()


# ---- BASELINE-size properties (no oracle at this size: size-independent invariants) --------------

def test_full_size_block_properties(dev):
    """cfg2 block shape [10 000, 512]: Gram symmetry/linearity, trace identity, and the eigen
    residual of the exact-mode result, checked with float64 torch matmuls on the device."""
    from ganspace_amd import ops
    from ganspace_amd.estimators import IPCAEstimator
    gen = torch.Generator(device="cpu").manual_seed(0)
    A = torch.randn(160, 512, generator=gen, dtype=torch.float64) * (1.04 ** -torch.arange(160.0, dtype=torch.float64))[:, None]
    est = IPCAEstimator(80, "exact")
    C = torch.zeros(512, 512, dtype=torch.float64, device=dev)
    s1 = torch.zeros(512, dtype=torch.float64, device=dev)
    n = 0
    for i in range(4):
        Z = torch.randn(10000, 160, generator=gen, dtype=torch.float64)
        X = (Z @ A + 0.05 * torch.randn(10000, 512, generator=gen, dtype=torch.float64) + 0.1).float().to(dev)
        assert est.fit_partial(X)
        Xd = X.double()
        C += Xd.T @ Xd
        s1 += Xd.sum(0)
        n += X.shape[0]
    mean = s1 / n
    Cc = C - n * torch.outer(mean, mean)
    t = est.transformer
    V = torch.from_numpy(t.components_).to(dev).double()
    lam = torch.from_numpy(t.singular_values_).to(dev) ** 2
    assert (V @ V.T - torch.eye(80, dtype=torch.float64, device=dev)).abs().max().item() < 1e-6
    resid = (Cc @ V.T - V.T * lam).abs().max().item()
    assert resid < 2e-5 * lam[0].item()
    np.testing.assert_allclose(t.mean_, mean.cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(t.var_ * n, torch.diag(Cc).cpu().numpy(), rtol=1e-5)
    assert abs(float(np.sum(t.explained_variance_ratio_)) - (lam.sum() / torch.trace(Cc)).item()) < 1e-6
    assert int(t.n_samples_seen_) == n


def test_nccl_single_rank_allreduce_of_estimator_state(dev):
    """The RCCL path with one rank (all a 1-GPU box can run): export -> two all-reduces -> re-centre
    (HIP kernel) -> import must leave the fitted result unchanged."""
    import torch.distributed as dist
    from ganspace_amd import distributed as D
    from ganspace_amd.estimators import IPCAEstimator
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        case = gin.IPCA_CASES["d512_k20"]
        est = IPCAEstimator(case["k"], "exact")
        for X in gin.ipca_blocks(case):
            est.fit_partial(torch.from_numpy(X).to(dev))
        before = est.transformer.components_.copy()
        sv = est.transformer.singular_values_.copy()
        D.allreduce_estimator(est)
        cos = O.signed_cosines(est.transformer.components_, before)
        assert cos.min() > 1 - 1e-9
        np.testing.assert_allclose(est.transformer.singular_values_, sv, rtol=2e-6)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("d,k,rows", [(2048, 32, 3000), (1000, 200, 2500)])
def test_larger_feature_dims_exact_and_faithful(dev, d, k, rows):
    """Gram-side path beyond the bench shape: more macro tiles than CUs per chunk (d = 2048) and k close
    to the subspace-solver limit (falls back to the full Jacobi solver when 2p > d)."""
    from ganspace_amd.estimators import IPCAEstimator
    rs = np.random.RandomState(d)
    A = rs.standard_normal((96, d)) * (1.07 ** -np.arange(96))[:, None]
    blocks = [(rs.standard_normal((rows, 96)) @ A + 0.02 * rs.standard_normal((rows, d)) + 0.5).astype(np.float32)
              for _ in range(2)]
    ex = O.exact_pca(blocks, k)
    est = IPCAEstimator(k, "exact")
    for X in blocks:
        assert est.fit_partial(torch.from_numpy(X).to(dev))
    r = 24
    cos = O.signed_cosines(est.transformer.components_[:r], ex["components_"][:r])
    assert cos.min() > 1 - 1e-6, cos.min()
    np.testing.assert_allclose(est.transformer.singular_values_[:r], ex["singular_values_"][:r], rtol=5e-5)
    fa = IPCAEstimator(k, "faithful")
    orc = O.IPCAEstimatorOracle(k, "gram")
    for X in blocks:
        assert fa.fit_partial(torch.from_numpy(X).to(dev))
        orc.fit_partial(X)
    cos = O.signed_cosines(fa.transformer.components_[:r], orc.transformer.components_[:r])
    assert cos.min() > 1 - 1e-6, cos.min()


def test_maximum_gram_side_dimension_smoke(dev):
    """d = 8192 is the largest Gram-side feature dim (slabs + accumulators ~6 GB): shapes, orthonormality, order."""
    from ganspace_amd.estimators import IPCAEstimator
    g = torch.Generator(device=dev).manual_seed(3)
    A = torch.randn(16, 8192, device=dev, generator=g) * (1.5 ** -torch.arange(16, device=dev))[:, None]
    est = IPCAEstimator(8, "exact")
    for _ in range(2):
        X = torch.randn(600, 16, device=dev, generator=g) @ A + 0.01 * torch.randn(600, 8192, device=dev, generator=g)
        assert est.fit_partial(X)
    comp, stdev, ratio = est.get_components()
    assert comp.shape == (8, 8192)
    G = comp.astype(np.float64) @ comp.astype(np.float64).T
    assert np.abs(G - np.eye(8)).max() < 1e-5
    assert np.all(np.diff(stdev) <= 0) and 0.9 < ratio.sum() <= 1.0 + 1e-6


def test_tiny_and_ragged_later_blocks(dev):
    """After the first block any number of rows (even 1) is a legal partial_fit batch in sklearn."""
    from ganspace_amd.estimators import get_estimator
    rs = np.random.RandomState(4)
    A = rs.standard_normal((10, 48))
    sizes = [64, 1, 7, 130, 2]
    blocks = [(rs.standard_normal((m, 10)) @ A + 0.1 * rs.standard_normal((m, 48))).astype(np.float32) for m in sizes]
    est = get_estimator("ipca", 5, 1.0)
    orc = O.IPCAEstimatorOracle(5, "svd")
    for X in blocks:
        assert est.fit_partial(torch.from_numpy(X).to(dev))
        orc.fit_partial(X)
    cos = O.signed_cosines(est.transformer.components_, orc.transformer.components_)
    assert cos.min() > 1 - 1e-5, cos
    np.testing.assert_allclose(est.transformer.singular_values_, orc.transformer.singular_values_, rtol=1e-4)
    assert int(est.transformer.n_samples_seen_) == sum(sizes)


def test_feature_count_mismatch_and_bad_k_raise(dev):
    from ganspace_amd.estimators import get_estimator
    est = get_estimator("ipca", 4, 1.0)
    assert est.fit_partial(torch.randn(16, 32, device=dev))
    with pytest.raises(ValueError):
        est.transformer.partial_fit(torch.randn(16, 40, device=dev))      # sklearn: feature-count change
    est2 = get_estimator("ipca", 64, 1.0)
    assert est2.fit_partial(torch.randn(128, 32, device=dev)) is False     # k > n_features -> ValueError -> False


@pytest.mark.parametrize("d,k,decay,probe", [(256, 40, 1.0, False), (256, 40, 3.0, False), (192, 21, 1.5, True),
                                              (512, 80, 1.0, True)])
def test_faithful_many_blocks_deferred_diagonalisation_matches_oracle(dev, d, k, decay, probe):
    """From the fifth block on the faithful mode carries (basis, k x k matrix) instead of diagonalising every block
    (gs_ipca.hip / invsub_iterate): the recurrence - sklearn's truncation to the k leading directions after every
    block - must come out the same as the oracle's block-by-block eigendecomposition, whether the components are
    only read at the end or also in the middle of the stream (which folds the pending rotation into the state)."""
    from ganspace_amd import _lib
    from ganspace_amd.estimators import get_estimator
    lib = _lib.load()
    rng = np.random.default_rng(7)
    basis = np.linalg.qr(rng.standard_normal((d, d)))[0] * np.sqrt((1.0 / np.arange(1, d + 1)) ** decay)
    blocks = [(rng.standard_normal((700, d)) @ basis.T + 0.25).astype(np.float32) for _ in range(16)]
    est = get_estimator("ipca", k, 1.0)
    orc = O.IPCAEstimatorOracle(k, "gram")
    carried = 0
    for i, X in enumerate(blocks):
        assert est.fit_partial(torch.from_numpy(X).to(dev)) is True
        orc.fit_partial(X)
        h = est.transformer._h
        carried += int(lib.gs_ipca_last_sweeps(h) == 0 and lib.gs_ipca_last_mults(h) > 0)
        if probe and i in (6, 11):
            mid = est.get_components()[0]
            sv = orc.transformer.singular_values_
            live = sv > 1e-3 * sv[0]
            assert O.signed_cosines(mid[live], orc.transformer.components_[live]).min() > 1 - 2e-6, i
    if decay <= 1.5:
        assert carried >= 8, carried          # blocks 5.. take the deferred path on a gently decaying spectrum
    comp = est.get_components()[0]
    sv, svo = est.transformer.singular_values_, orc.transformer.singular_values_
    live = svo > 1e-3 * svo[0]
    cos = O.signed_cosines(comp[live], orc.transformer.components_[live])
    assert cos.min() > 1 - 2e-6, cos.min()
    np.testing.assert_allclose(sv[live], svo[live], rtol=1e-4)
    np.testing.assert_allclose(sv, svo, atol=5e-4 * svo[0])
    np.testing.assert_allclose(est.transformer.explained_variance_ratio_[live],
                               orc.transformer.explained_variance_ratio_[live], rtol=2e-4)
    np.testing.assert_allclose(est.transformer.mean_, orc.transformer.mean_, atol=2e-6)


@pytest.mark.parametrize("precision,cos_tol", [("f32", 3e-6), ("bf16x6", 8e-6)])
@pytest.mark.parametrize("d,k,rows,decay,probe", [(6000, 24, 400, 1.0, False), (3000, 21, 300, 2.0, True)])
def test_smallside_many_blocks_deferred_diagonalisation_matches_oracle(dev, d, k, rows, decay, probe, precision, cos_tol):
    """Small side (T = M M^T) with the diagonalisation deferred: from the fifth block on the state is W = Q^T M
    (gs_smallside.hip).  Checked against the sklearn-recurrence oracle (SVD of the stacked matrix), read at the end
    and, in the probing variant, in mid-stream (which folds the pending rotation back into unit components)."""
    from ganspace_amd import _lib
    from ganspace_amd.estimators import IPCAEstimator
    lib = _lib.load()
    rng = np.random.default_rng(11)
    latent = 96
    A = rng.standard_normal((latent, d)) * np.sqrt((1.0 / np.arange(1, latent + 1)) ** decay)[:, None]
    blocks = [(rng.standard_normal((rows, latent)) @ A + 0.02 * rng.standard_normal((rows, d)) + 0.1).astype(np.float32)
              for _ in range(14)]
    est = IPCAEstimator(k, "smallside", precision=precision)
    orc = O.IPCAEstimatorOracle(k, "svd")
    carried = 0
    for i, X in enumerate(blocks):
        assert est.fit_partial(torch.from_numpy(X).to(dev)) is True
        orc.fit_partial(X)
        h = est.transformer._h
        carried += int(i >= 4 and lib.gs_ipca_last_sweeps(h) == 0 and lib.gs_ipca_last_mults(h) > 0)
        if probe and i in (6, 10):
            mid = est.get_components()[0]
            assert O.signed_cosines(mid, orc.transformer.components_).min() > 1 - cos_tol, i
    if decay <= 1.0:
        assert carried >= 6, carried
    comp = est.get_components()[0]
    cos = O.signed_cosines(comp, orc.transformer.components_)
    assert cos.min() > 1 - cos_tol, cos.min()
    np.testing.assert_allclose(est.transformer.singular_values_, orc.transformer.singular_values_, rtol=2e-4)
    np.testing.assert_allclose(est.transformer.explained_variance_ratio_, orc.transformer.explained_variance_ratio_,
                               rtol=4e-4)
    np.testing.assert_allclose(est.transformer.mean_, orc.transformer.mean_, atol=2e-6)
    np.testing.assert_allclose(est.transformer.var_, orc.transformer.var_, rtol=1e-4)


@pytest.mark.parametrize("case", ["shift", "lowrank", "tiny_late_blocks"])
def test_faithful_deferred_path_falls_back_when_its_assumptions_break(dev, case):
    """The carried-basis blocks assume a (t + 1)-fold gap behind lambda_k.  A distribution shift in mid-stream (new
    directions 30x stronger than anything seen), data of rank < k (lambda_k = 0: dead pivots) and late blocks of a
    handful of rows must all still reproduce the oracle's recurrence - through the retry / Rayleigh-Ritz fallback."""
    from ganspace_amd.estimators import get_estimator
    rng = np.random.default_rng(3)
    d, k = 160, 16
    Qb = np.linalg.qr(rng.standard_normal((d, d)))[0]
    def draw(rows, cols, scale):
        return (rng.standard_normal((rows, len(cols))) * scale) @ Qb[:, cols].T
    blocks = []
    for i in range(14):
        if case == "shift":
            X = draw(500, np.arange(0, 40), 1.0 / np.arange(1, 41)) if i < 8 else \
                draw(500, np.arange(0, 40), 1.0 / np.arange(1, 41)) + draw(500, np.arange(60, 90), 30.0 / np.arange(1, 31))
        elif case == "lowrank":
            X = draw(400, np.arange(0, 10), 1.0 / np.arange(1, 11))
        else:
            X = draw(600 if i < 6 else 7 + i, np.arange(0, 60), 1.0 / np.arange(1, 61))
        blocks.append((X + 0.5).astype(np.float32))
    est = get_estimator("ipca", k, 1.0)
    orc = O.IPCAEstimatorOracle(k, "gram")
    for X in blocks:
        assert est.fit_partial(torch.from_numpy(X).to(dev)) is True
        orc.fit_partial(X)
    comp = est.get_components()[0]
    svo = orc.transformer.singular_values_
    live = svo > 1e-3 * svo[0]
    assert live.sum() >= (10 if case == "lowrank" else k)
    cos = O.signed_cosines(comp[live], orc.transformer.components_[live])
    assert cos.min() > 1 - 3e-6, (case, cos.min())
    np.testing.assert_allclose(est.transformer.singular_values_[live], svo[live], rtol=1e-4)
    np.testing.assert_allclose(est.transformer.singular_values_, svo, atol=5e-4 * svo[0])
    np.testing.assert_allclose(est.transformer.mean_, orc.transformer.mean_, atol=2e-6)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_exact_mode_resident_rows_are_merged_and_match_block_by_block(dev, precision):
    """fit_partial(X, resident=True): the exact mode may postpone the contraction of rows the caller keeps alive and
    merge contiguous calls into launches of 131 072 rows (d = 512: the wide kernels).  Same statistics as feeding the
    blocks one by one - also when a call is not the continuation of the previous one, when resident and ordinary
    calls alternate, and when the state is exported in between."""
    from ganspace_amd.estimators import IPCAEstimator
    rng = np.random.default_rng(5)
    d, k = 512, 24
    A = rng.standard_normal((64, d)) * (1.25 ** -np.arange(64))[:, None]
    big = torch.from_numpy((rng.standard_normal((300000, 64)) @ A + 0.05 * rng.standard_normal((300000, d)) + 0.2)
                           .astype(np.float32)).to(dev)
    pieces = [(0, 50000), (50000, 120000), (120000, 170000),      # contiguous: one 131 072-row launch + a rest
              (200000, 260000),                                     # gap: not a continuation
              (170000, 200000)]                                     # ordinary call in between
    a = IPCAEstimator(k, "exact", precision=precision)
    b = IPCAEstimator(k, "exact", precision=precision)
    for i, (lo, hi) in enumerate(pieces):
        assert a.fit_partial(big[lo:hi], resident=(i != 4))
        assert b.fit_partial(big[lo:hi].clone())
        if i == 2:
            st = a.transformer.export_state()               # flushes what is pending
            assert float(st[0].item()) == 170000.0
    assert a.fit_partial(big[260000:300000], resident=True)   # still pending when the components are read
    assert b.fit_partial(big[260000:300000].clone())
    ca, cb = a.get_components()[0], b.get_components()[0]
    cos = np.abs(O.signed_cosines(ca, cb))
    assert cos.min() > 1 - (1e-6 if precision == "f32" else 2e-5), cos.min()
    np.testing.assert_allclose(a.transformer.singular_values_, b.transformer.singular_values_, rtol=2e-6 if precision == "f32" else 1e-4)
    np.testing.assert_allclose(a.transformer.mean_, b.transformer.mean_, atol=1e-6)
    np.testing.assert_allclose(a.transformer.var_, b.transformer.var_, rtol=1e-5)
    assert int(a.transformer.n_samples_seen_) == 300000
    # against float64 on the host
    Xh = big.cpu().numpy().astype(np.float64)
    order = [r for lo, hi in pieces for r in range(lo, hi)] + list(range(260000, 300000))
    Xc = Xh[order] - Xh[order].mean(0)
    w, V = np.linalg.eigh(Xc.T @ Xc)
    ref = V[:, ::-1][:, :k].T
    assert np.abs(O.signed_cosines(ca, ref)).min() > 1 - (2e-6 if precision == "f32" else 3e-5)


def test_plain_bf16_contraction_keeps_the_leading_components(dev):
    """``precision="bf16"`` (the single-pass contraction that makes the Gram update HBM-bound): the
    rounding errors of the rows are independent, so a Gram entry summed over n rows is off by ~2^-9 / sqrt(n) of its
    Cauchy-Schwarz scale - measured here - and the leading components stay inside the north_star's tolerance
    (top-20 cosine >= 0.999) with three digits to spare on cfg2-shaped data (d = 512, k = 80, 131 072 + rows through
    the wide kernel and a tail through the tiled one)."""
    from ganspace_amd import ops
    from ganspace_amd.estimators import IPCAEstimator
    rng = np.random.default_rng(17)
    d, k, n = 512, 80, 300_000
    A = rng.standard_normal((160, d)) * (1.04 ** -np.arange(160))[:, None]
    X = (rng.standard_normal((n, 160)) @ A + 0.05 * rng.standard_normal((n, d)) + 0.2).astype(np.float32)
    Xd = torch.from_numpy(X).to(dev)
    est = IPCAEstimator(k, "exact", precision="bf16")
    ref = IPCAEstimator(k, "exact")
    for lo in range(0, n, 50_000):
        assert est.fit_partial(Xd[lo:lo + 50_000], resident=True)
        assert ref.fit_partial(Xd[lo:lo + 50_000], resident=True)
    cos = O.signed_cosines(est.get_components()[0], ref.get_components()[0])
    assert cos[:20].min() > 1 - 1e-6, cos[:20]            # north_star asks for >= 0.999
    assert np.abs(cos).min() > 0.999, np.abs(cos).min()   # all 80: eigenvalue gaps of a few per cent
    np.testing.assert_allclose(est.transformer.singular_values_, ref.transformer.singular_values_, rtol=2e-4)
    np.testing.assert_allclose(est.transformer.mean_, ref.transformer.mean_, atol=1e-6)
    # error of the Gram itself: ~2^-9 / sqrt(rows), far below the 4e-3 bound of a single product
    G, _ = ops.gram_accumulate(Xd[:131072], precision="bf16")
    G32, _ = ops.gram_accumulate(Xd[:131072])
    scale = torch.sqrt(torch.outer(torch.diag(G32), torch.diag(G32)))
    assert float(((G - G32) / scale).abs().max()) < 2e-4


@pytest.mark.parametrize("rows", [1_000_000, 1_048_576 + 4_099])
def test_plain_bf16_contraction_at_the_benchmarked_launch_size(dev, rows):
    """The launch ``bench.py`` prices for ``roofline_hbm``: all the resident rows of the cfg2 job in ONE
    ``gram_bf16_glds_kernel`` launch (8192-row chunks, 2^20 rows at most; a second launch takes the rest) against a
    float64 contraction of the same device rows (torch float64 matmul as the checker).  Column sums come from the
    float32 rows, not from their bf16 images: float32-exact."""
    from ganspace_amd import ops
    g = torch.Generator(device=dev).manual_seed(rows)
    d = 512
    X = torch.randn(rows, d, device=dev, generator=g) * torch.linspace(0.3, 2.5, d, device=dev) + 0.4
    shift = X[:4096].mean(0) + 0.01
    G, cs = ops.gram_accumulate(X, shift=shift, precision="bf16")
    Gref = torch.zeros(d, d, dtype=torch.float64, device=dev)
    csref = torch.zeros(d, dtype=torch.float64, device=dev)
    absum = torch.zeros(d, dtype=torch.float64, device=dev)
    for lo in range(0, rows, 131072):
        Xc = X[lo:lo + 131072].double() - shift.double()
        Gref += Xc.T @ Xc
        csref += Xc.sum(0)
        absum += Xc.abs().sum(0)
    scale = torch.sqrt(torch.outer(torch.diag(Gref), torch.diag(Gref)))
    err = float(((G.double() - Gref) / scale).abs().max())
    assert err < 1e-4, err                                  # ~2^-9 / sqrt(rows) per entry; 4e-3 is the per-product bound
    assert float((cs.double() - csref).abs().max()) <= 2e-6 * float(absum.max())
    assert float((G - G.T).abs().max()) == 0.0


def test_second_stream_fold_equals_in_launch_fold(tmp_path):
    """Wide launches (d = 512, >= 20 000 rows) fold their float32 slabs into the float64 accumulators on a second stream
    while the next launch computes (``gram_fold_light_kernel``, ordered by events); ``GS_GRAM_NO_AUX_FOLD=1`` (measurement build)
    keeps the folds on the spare workgroups of the next launch.  Same slabs, same float64 additions per element: faithful and
    exact estimators on 30 000-row blocks agree to rounding (``tools/aux_fold_check.py``, one process per variant -
    the switch is read when the workspace is created)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # (round 4: the switch is compiled out of the production library - the in-launch variant runs on the measurement build,
    #  ganspace_amd/lib_measure, same sources with -DGS_MEASURE_BUILD)
    from ganspace_amd import _build
    measure_lib = _build.build(measure=True, verbose=False)
    for tag, extra in (("aux", {}), ("inline", {"GS_GRAM_NO_AUX_FOLD": "1", "GANSPACE_HIP_LIB": measure_lib})):
        path = str(tmp_path / f"{tag}.npy")
        env = {k: v for k, v in os.environ.items() if k not in ("GS_GRAM_NO_AUX_FOLD", "GANSPACE_HIP_LIB")}
        env.update(extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "aux_fold_check.py"), path], cwd=root, env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(np.load(path))
    a, b = outs
    assert a.shape == b.shape and np.isfinite(a).all()
    assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(a).max())

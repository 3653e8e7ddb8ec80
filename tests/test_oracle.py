"""The oracle against the reference-generated golden fixtures and sklearn's own
known-answer tests (CPU only).  Pins oracle/ before anything trusts it."""
import os

import numpy as np
import pytest

import inputs as gin
from oracle import ipca as O
from oracle import synth, zstream


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"ipca_ref_{name}.npz"), allow_pickle=False)


def _fit(cls_or_kind, case):
    est = O.IPCAEstimatorOracle(case["k"], cls_or_kind)
    svs = []
    for X in gin.ipca_blocks(case):
        assert est.fit_partial(X)
        svs.append(np.array(est.transformer.singular_values_, dtype=np.float64))
    return est, np.stack(svs)


def _n_meaningful(case):
    # components past the data rank / at the noise floor are ill conditioned
    return case["ncheck"]


@pytest.mark.parametrize("name", list(gin.IPCA_CASES))
@pytest.mark.parametrize("kind", ["svd", "gram"])
def test_recurrence_oracles_match_reference(golden_dir, name, kind):
    case = gin.IPCA_CASES[name]
    g = _load(golden_dir, name)
    est, svs = _fit(kind, case)
    t = est.transformer
    r = _n_meaningful(case)
    comp, stdev, ratio = est.get_components()
    # signed cosine: sign convention (svd_flip) is part of the contract
    cos = O.signed_cosines(comp[:r], g["components"][:r])
    tol = 2e-6 if kind == "svd" else 2e-5   # block 1 of the reference runs a float32 SVD
    assert cos.min() > 1 - tol, (name, kind, cos.min())
    scale = g["singular_values"][0]
    np.testing.assert_allclose(t.singular_values_[:r], g["singular_values"][:r], rtol=5e-5)
    np.testing.assert_allclose(t.singular_values_, g["singular_values"], atol=2e-5 * scale)
    np.testing.assert_allclose(svs, g["per_block_singular_values"], atol=2e-5 * scale)
    np.testing.assert_allclose(t.mean_, g["mean"], rtol=0, atol=1e-6 * max(1, np.abs(g["mean"]).max()))
    np.testing.assert_allclose(t.var_, g["var"], rtol=2e-5)
    np.testing.assert_allclose(stdev[:r], g["stdev"][:r], rtol=5e-5)
    np.testing.assert_allclose(ratio[:r], g["var_ratio"][:r], rtol=1e-4)
    assert int(t.n_samples_seen_) == int(g["n_samples_seen"]) == sum(case["blocks"])
    assert est.get_param_str() == str(g["param_str"])


def test_exact_pca_equals_ipca_when_k_equals_d(golden_dir):
    case = gin.IPCA_CASES["d200_k200_ragged"]
    g = _load(golden_dir, "d200_k200_ragged")
    ex = O.exact_pca(gin.ipca_blocks(case), case["k"])
    # well separated leading part of the spectrum; the tail is float32-noise dominated
    r = case["ncheck"]
    cos = O.signed_cosines(ex["components_"][:r], g["components"][:r])
    assert cos.min() > 1 - 1e-5
    np.testing.assert_allclose(ex["singular_values_"][:r], g["singular_values"][:r], rtol=2e-5)
    np.testing.assert_allclose(ex["mean_"], g["mean"], atol=1e-6)


def test_exact_pca_top_components_match_truncated_ipca(golden_dir):
    # k=20 of d=512: exact PCA agrees with the truncated recurrence on the leading components
    case = gin.IPCA_CASES["d512_k20"]
    g = _load(golden_dir, "d512_k20")
    ex = O.exact_pca(gin.ipca_blocks(case), case["k"])
    cos = np.abs(O.signed_cosines(ex["components_"][:10], g["components"][:10]))
    assert cos.min() > 0.999


def test_first_block_smaller_than_k_returns_false(capsys):
    est = O.IPCAEstimatorOracle(20, "svd")
    assert est.fit_partial(np.zeros((10, 32), dtype=np.float32)) is False
    assert "IPCA error" in capsys.readouterr().out


# ---- scikit-learn's own known-answer checks, re-run on the restatement -------------
# (sklearn/decomposition/tests/test_incremental_pca.py: test_singular_values :351,
#  test_incremental_pca_against_pca_iris :309, test_incremental_pca_partial_fit :289)

def _fit_in_batches(est, X, bs):
    for i in range(0, X.shape[0], bs):
        est.partial_fit(X[i:i + bs])
    return est


@pytest.mark.parametrize("cls", [O.SklearnRecurrenceOracle, O.GramRecurrenceOracle])
def test_known_singular_values(cls):
    from sklearn import datasets
    from sklearn.decomposition import PCA
    rng = np.random.RandomState(0)
    X = datasets.make_low_rank_matrix(100, 110, tail_strength=0.0, effective_rank=3, random_state=rng)
    pca = PCA(n_components=3, svd_solver="full")
    Xp = pca.fit_transform(X)
    Xp /= np.sqrt(np.sum(Xp ** 2.0, axis=0))
    Xp[:, 0] *= 3.142
    Xp[:, 1] *= 2.718
    X_hat = Xp @ pca.components_
    est = cls(3).partial_fit(X_hat)
    np.testing.assert_allclose(est.singular_values_, [3.142, 2.718, 1.0], rtol=0, atol=1e-12)


@pytest.mark.parametrize("cls", [O.SklearnRecurrenceOracle, O.GramRecurrenceOracle])
def test_against_sklearn_on_iris(cls):
    from sklearn import datasets
    from sklearn.decomposition import IncrementalPCA
    X = datasets.load_iris().data
    ref = IncrementalPCA(n_components=2, batch_size=25).fit(X)
    est = _fit_in_batches(cls(2), X, 25)
    assert O.signed_cosines(est.components_, ref.components_).min() > 1 - 1e-10
    np.testing.assert_allclose(est.singular_values_, ref.singular_values_, rtol=1e-9)
    np.testing.assert_allclose(est.explained_variance_ratio_, ref.explained_variance_ratio_, rtol=1e-9)
    np.testing.assert_allclose(est.mean_, ref.mean_, rtol=1e-12)
    np.testing.assert_allclose(est.var_, ref.var_, rtol=1e-10)


def test_batch_signs_stable():
    # test_incremental_pca_batch_signs :220 - signs must not depend on the batch size
    rng = np.random.RandomState(1999)
    X = rng.randn(100, 3)
    comps = [_fit_in_batches(O.GramRecurrenceOracle(3), X, bs).components_ for bs in (10, 20, 50)]
    for a, b in zip(comps[:-1], comps[1:]):
        np.testing.assert_allclose(np.sign(a), np.sign(b))


# ---- z-stream protocol (SURVEY §A.3 probe values) ----------------------------------

def test_zstream_seed_parity_smoke():
    assert zstream.batch_seeds(3) == [1791095845, 2135392491, 946286476]
    z = zstream.stylegan_z_batch(1791095845, 4, 512)
    np.testing.assert_allclose(z[0, :4], [0.76455638, -1.12429114, -0.13731647, 0.52814697], rtol=1e-6)


def test_loop_plan_matches_baseline_configs():
    # cfg1: -n=10_000 -b=512 -c=20  -> N=9728, NB=2000, 5 blocks
    B, N, NB, n_lat, nb = zstream.loop_plan(10_000, 512, 20)
    assert (B, N, NB, nb) == (512, 9728, 2000, 5) and n_lat == ((N + NB - 1) // B + 1) * B
    # cfg2: -n=1_000_000 -b=10_000 -c=80 -> 100 blocks of 10 000, 1 010 000 latent rows
    B, N, NB, n_lat, nb = zstream.loop_plan(1_000_000, 10_000, 80)
    assert (N, NB, nb, n_lat) == (1_000_000, 10_000, 100, 1_010_000)


def test_iter_blocks_truncates_tail_minibatch():
    lat = np.arange(3072 * 4, dtype=np.float32).reshape(3072, 4)
    blocks = list(zstream.iter_blocks(lat, 1024, 512, 4))   # N=1024, NB=2000
    assert len(blocks) == 1 and blocks[0].shape == (2000, 4)
    np.testing.assert_array_equal(blocks[0], lat[:2000])


# ---- mapping network oracle vs the in-tree reference G_mapping ----------------------

def test_mapping_oracle_matches_reference_gmapping(golden_dir):
    g = np.load(os.path.join(golden_dir, "mapping_gmapping_ref.npz"))
    W, b = gin.mapping_weights()
    w = synth.mapping_network(gin.mapping_z(), W, b, lr_mul=gin.MAPPING_CASE["lr_mul"])
    np.testing.assert_allclose(w, g["w_f64"], rtol=1e-9, atol=1e-11)
    scale = np.abs(g["w_f64"]).max()
    assert np.abs(w - g["w_f32"]).max() < 5e-5 * scale


def test_parallel_z_generation_is_bit_identical_to_the_serial_protocol(monkeypatch):
    from ganspace_amd import _zgen
    seeds = zstream.batch_seeds(6)
    serial = [zstream.stylegan_z_batch(s, 64, 512) for s in seeds]
    monkeypatch.setenv("GANSPACE_ZGEN_WORKERS", "3")       # force the process pool
    pooled = list(_zgen.generate("stylegan", seeds, 64, 512))
    for a, b in zip(serial, pooled):
        np.testing.assert_array_equal(a, b)
    big = list(_zgen.generate("biggan", seeds[:4], 16, 128, 1.0))
    for s, b in zip(seeds[:4], big):
        np.testing.assert_array_equal(zstream.biggan_z_batch(s, 16), b)


def test_smallside_torch_oracle_matches_sklearn_recurrence_oracle():
    """oracle/smallside_torch.py (float64, r x r side, plain torch matmuls - the checker used at the benchmarked
    wide-feature shapes) restates the same recurrence as the SVD-form oracle."""
    torch = pytest.importorskip("torch")
    from oracle.smallside_torch import SmallSideTorchOracle, lowrank_plus_noise_blocks
    k = 10
    a, b = SmallSideTorchOracle(k), O.SklearnRecurrenceOracle(k)
    for X in lowrank_plus_noise_blocks(600, 6, rows=120, latent=24, decay=1.15, seed=3):
        a.partial_fit(X)
        b.partial_fit(X.numpy().astype(np.float64))
    assert O.signed_cosines(a.components_, b.components_).min() > 1 - 1e-10
    np.testing.assert_allclose(a.singular_values_, b.singular_values_, rtol=1e-10)
    np.testing.assert_allclose(a.mean_, b.mean_, atol=1e-12)
    np.testing.assert_allclose(a.var_, b.var_, rtol=1e-10)
    np.testing.assert_allclose(a.explained_variance_ratio_, b.explained_variance_ratio_, rtol=1e-10)
    assert a.n_samples_seen_ == b.n_samples_seen_ == 720


@pytest.mark.parametrize("m,n", [(3000, 200), (150, 900), (60, 40)])
def test_fbpca_port_recovers_leading_singular_triplets(m, n):
    """oracle/fbpca_port.py (restatement of fbpca.pca, raw=True; the package itself is absent - parity unpinned by the
    reference): on a matrix with a decaying spectrum the n_iter = 2, l = 2k range finder reproduces the leading
    singular values / right singular vectors of the dense SVD; both branches (m >= n, m < n) and the dense fall-back."""
    from oracle import fbpca_port
    rs = np.random.RandomState(4)
    k = 8
    r = min(m, n)
    U0 = np.linalg.qr(rs.standard_normal((m, r)))[0]
    V0 = np.linalg.qr(rs.standard_normal((n, r)))[0]
    sv = 10.0 * 1.35 ** -np.arange(r)
    A = (U0 * sv) @ V0.T
    np.random.seed(11)
    U, s, Va = fbpca_port.pca(A, k=k, raw=True, n_iter=2, l=2 * k)
    assert U.shape == (m, k) and Va.shape == (k, n)
    np.testing.assert_allclose(s, sv[:k], rtol=1e-6)
    cos = np.abs(np.sum(Va * V0[:, :k].T, axis=1))
    assert cos.min() > 1 - 1e-8
    # deterministic in the global stream, like fbpca
    np.random.seed(11)
    _, s2, Va2 = fbpca_port.pca(A, k=k, raw=True, n_iter=2, l=2 * k)
    np.testing.assert_array_equal(Va, Va2)

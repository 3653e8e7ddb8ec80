"""Whole-matrix estimators of the reference (``estimators.py:84-160``) on the device beyond the Gram-side shapes:
``fbpca`` as the randomized range finder it is (both branches of fbpca.pca: rows >= feat_dim and rows < feat_dim),
``pca`` for feat_dim > 8192 from the small side, and the building blocks ``gs_column_moments`` / ``gs_randomized_pca``.

fbpca itself is not installed and the reference holds no output of it (parity unpinned by the reference): the checker
is ``oracle/fbpca_port.py``, a restatement of fbpca.pca's published algorithm, fed the SAME test matrix (same state of
NumPy's global stream, which is where fbpca draws it)."""
import ctypes as C

import numpy as np
import pytest

from oracle import fbpca_port

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    return torch.device("cuda", 0)


def _matrix(n, d, latent, seed, decay=1.3, noise=0.02, offset=0.4):
    rs = np.random.RandomState(seed)
    A = rs.standard_normal((latent, d)) * (decay ** -np.arange(latent))[:, None] * 3.0
    return (rs.standard_normal((n, latent)) @ A + noise * rs.standard_normal((n, d)) + offset).astype(np.float32)


@pytest.mark.parametrize("n,d,k", [(20000, 512, 20),        # rows >= feat_dim, the W-space shape
                                   (3001, 200, 12),         # ragged row count, feat_dim not a multiple of the tile
                                   (600, 9000, 10),         # rows < feat_dim: fbpca's second branch, feat_dim > 8192
                                   (12000, 8704, 16)])      # rows >= feat_dim > 8192: no d x d Gram exists for this one
def test_fbpca_estimator_matches_restatement_on_the_same_test_matrix(dev, n, d, k):
    from ganspace_amd.estimators import get_estimator
    X = _matrix(n, d, 40, seed=n + d)
    X -= X.mean(axis=0, keepdims=True, dtype=np.float32)          # compute() centres before fit (decomposition.py:279-281)
    est = get_estimator("fbpca", k, 1.0)
    assert est.get_param_str() == f"fbpca_c{k}_it2_l{2 * k}" and est.batch_support is False
    np.random.seed(5)
    est.fit(torch.from_numpy(X).to(dev))
    state_after = np.random.get_state()[1][:4].copy()
    orc = fbpca_port.FacebookPCAEstimatorOracle(k)
    np.random.seed(5)
    orc.fit(X.astype(np.float64))
    assert np.array_equal(np.random.get_state()[1][:4], state_after)     # both consumed the stream identically
    comp, stdev, ratio = est.get_components()
    ocomp, ostdev, oratio = orc.get_components()
    assert comp.shape == (k, d) and comp.dtype == np.float32
    acos = np.abs(np.sum(comp.astype(np.float64) * ocomp, axis=1))
    assert acos.min() > 1 - 1e-5, acos
    np.testing.assert_allclose(stdev, ostdev, rtol=2e-4)
    np.testing.assert_allclose(ratio, oratio, rtol=4e-4)
    np.testing.assert_allclose(np.asarray(est.transformer.mean_).ravel(), X.astype(np.float64).mean(0), atol=1e-6)
    gram = comp.astype(np.float64) @ comp.astype(np.float64).T
    assert np.abs(gram - np.eye(k)).max() < 1e-5


@pytest.mark.parametrize("n,d,k,latent,decay,noise", [
    (5000, 400, 10, 15, 1.3, 0.0),        # rank(A) = 15 (+ float32 rounding) < l = 20: Cholesky pivots die
    (6000, 200, 10, 40, 1.7, 0.02),       # s_1 / s_l ~ 2e3: cond(A^T A Q) ~ 4e6, one CholeskyQR pass loses ~1e-3
    (300, 2000, 8, 10, 1.5, 0.0)])        # rows < feat_dim branch, rank 10 < l = 16
def test_fbpca_low_rank_and_fast_decay(dev, n, d, k, latent, decay, noise):
    """The float64 CholeskyQR of the range-finder bases (twice per orthonormalisation) against the restatement's
    pivoted LU / QR where the iterate is rank deficient or badly conditioned."""
    from ganspace_amd.estimators import get_estimator
    X = _matrix(n, d, latent, seed=7 * n + d, decay=decay, noise=noise, offset=0.0)
    X -= X.mean(axis=0, keepdims=True, dtype=np.float32)
    est = get_estimator("fbpca", k, 1.0)
    np.random.seed(11)
    est.fit(torch.from_numpy(X).to(dev))
    orc = fbpca_port.FacebookPCAEstimatorOracle(k)
    np.random.seed(11)
    orc.fit(X.astype(np.float64))
    comp, stdev, _ = est.get_components()
    ocomp, ostdev, _ = orc.get_components()
    assert np.isfinite(comp).all() and np.isfinite(stdev).all()
    acos = np.abs(np.sum(comp.astype(np.float64) * ocomp, axis=1))
    assert acos.min() > 1 - 1e-5, acos
    np.testing.assert_allclose(stdev, ostdev, rtol=5e-4)
    gram = comp.astype(np.float64) @ comp.astype(np.float64).T
    assert np.abs(gram - np.eye(k)).max() < 1e-5


def test_fbpca_dense_fallback_for_small_matrices(dev):
    """fbpca hands matrices with l >= m / 1.25 or l >= n / 1.25 to a dense SVD (no randomness consumed)."""
    from ganspace_amd.estimators import get_estimator
    X = _matrix(4000, 64, 30, seed=2)
    est = get_estimator("fbpca", 30, 1.0)           # l = 60 >= 64 / 1.25
    before = np.random.get_state()[1][:4].copy()
    est.fit(torch.from_numpy(X).to(dev))
    assert np.array_equal(np.random.get_state()[1][:4], before)
    comp = est.get_components()[0]
    _, _, Vt = np.linalg.svd(X.astype(np.float64), full_matrices=False)
    proj = np.dot(Vt[:30], X.astype(np.float64).T).std(axis=1)
    order = np.argsort(proj)[::-1]
    assert np.abs(np.sum(comp[:12].astype(np.float64) * Vt[:30][order][:12], axis=1)).min() > 1 - 1e-6


def test_pca_estimator_beyond_the_gram_side(dev):
    """``--est=pca`` on a wide layer (feat_dim = 9000 > 8192): exact PCA from the rows x rows side."""
    from sklearn.decomposition import PCA
    from ganspace_amd.estimators import get_estimator
    X = _matrix(700, 9000, 30, seed=8)
    k = 10
    est = get_estimator("pca", k, 1.0)
    est.fit(torch.from_numpy(X).to(dev))
    comp, stdev, ratio = est.get_components()
    X64 = X.astype(np.float64)
    ref = PCA(k, svd_solver="full").fit(X64)
    ref_stdev = np.dot(ref.components_, X64.T).std(axis=1)
    acos = np.abs(np.sum(comp.astype(np.float64) * ref.components_, axis=1))
    assert acos.min() > 1 - 3e-6, acos
    np.testing.assert_allclose(stdev, ref_stdev, rtol=1e-4)
    np.testing.assert_allclose(ratio, ref_stdev ** 2 / X64.var(axis=0).sum(), rtol=2e-4)
    np.testing.assert_allclose(np.asarray(est.transformer.mean_).ravel(), X64.mean(0), atol=1e-5)


@pytest.mark.parametrize("rows,d,ld_extra,use_shift", [(1, 4, 0, False), (1000, 512, 0, True), (4097, 100, 28, True),
                                                       (70000, 36, 0, False)])
def test_column_moments_match_float64(dev, rows, d, ld_extra, use_shift):
    from ganspace_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(rows + d)
    Xfull = (rs.standard_normal((rows, d + ld_extra)) * 2 + 30.0).astype(np.float32)
    Xd = torch.from_numpy(Xfull).to(dev)[:, :d]
    shift = torch.from_numpy(Xfull[:, :d].astype(np.float64).mean(0) + 0.1).to(dev) if use_shift else None
    acc = torch.zeros((2, d), dtype=torch.float64, device=dev)
    acc[0] += 1.0                                        # accumulates
    _lib.check(lib.gs_column_moments(C.c_void_p(Xd.data_ptr()), rows, Xd.stride(0), d,
                                     C.c_void_p(shift.data_ptr() if use_shift else 0), C.c_void_p(acc[0].data_ptr()),
                                     C.c_void_p(acc[1].data_ptr()), _lib.current_stream_ptr()))
    X64 = Xfull[:, :d].astype(np.float64) - (shift.cpu().numpy() if use_shift else 0.0)
    np.testing.assert_allclose(acc[0].cpu().numpy(), 1.0 + X64.sum(0), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(acc[1].cpu().numpy(), (X64 * X64).sum(0), rtol=1e-12)


@pytest.mark.parametrize("name,n,d,k", [("pca", 2000, 203, 8), ("fbpca", 3001, 201, 12), ("fbpca", 150, 333, 6)])
def test_whole_matrix_estimators_accept_any_feature_width(dev, name, n, d, k):
    """The reference's PCAEstimator / FacebookPCAEstimator take any width (estimators.py:84-160); the device kernels read
    float4 rows, so widths that are not a multiple of 4 are zero-padded (a constant-zero feature changes nothing).
    Round 3 raised here (advisor finding)."""
    from ganspace_amd.estimators import get_estimator
    X = _matrix(n, d, 30, seed=n + d)
    X -= X.mean(axis=0, keepdims=True, dtype=np.float32)
    est = get_estimator(name, k, 1.0)
    np.random.seed(7)
    est.fit(torch.from_numpy(X).to(dev))
    comp, stdev, ratio = est.get_components()
    assert comp.shape == (k, d) and np.asarray(est.transformer.mean_).shape == (1, d)
    if name == "pca":
        from sklearn.decomposition import PCA
        ref = PCA(k, svd_solver="full").fit(X.astype(np.float64))
        acos = np.abs(np.sum(comp.astype(np.float64) * ref.components_, axis=1))
        assert acos.min() > 1 - 1e-6, acos
        np.testing.assert_allclose(stdev, np.sqrt(ref.explained_variance_ * (n - 1) / n), rtol=1e-5)
    else:
        orc = fbpca_port.FacebookPCAEstimatorOracle(k)
        np.random.seed(7)
        orc.fit(X.astype(np.float64))
        ocomp, ostdev, _ = orc.get_components()
        acos = np.abs(np.sum(comp.astype(np.float64) * ocomp, axis=1))
        assert acos.min() > 1 - 1e-5, acos
        np.testing.assert_allclose(stdev, ostdev, rtol=2e-4)

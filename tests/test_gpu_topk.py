"""GPU tests of the latency-first top-k eigensolver (gs_topk.hip) and its single-workgroup building blocks,
against NumPy/LAPACK float64 on the same matrices.  The solver stands in for LAPACK gesdd inside
``IncrementalPCA.partial_fit`` (sklearn/decomposition/_incremental_pca.py:362); end-to-end parity with the
reference is covered by test_gpu_parity.py / test_gpu_decomposition.py."""
import numpy as np
import pytest

import inputs as gin

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; the product path has no CPU fallback")
    from ganspace_amd import _lib
    _lib.load()
    return torch.device("cuda", 0)


def _spd(p, seed, cond=1e3):
    rs = np.random.RandomState(seed)
    Q, _ = np.linalg.qr(rs.standard_normal((p, p)))
    ev = np.logspace(0, -np.log10(cond), p)
    return (Q * ev) @ Q.T


@pytest.mark.parametrize("n,p,cond", [(512, 128, 1e3), (512, 128, 3e6), (2081, 128, 1e4), (512, 96, 1e5), (313, 64, 1e2),
                                       (200, 37, 1e4), (40, 5, 10.0), (9, 1, 1.0)])
def test_cholqr_matches_lapack(dev, n, p, cond):
    from ganspace_amd import ops
    rs = np.random.RandomState(100 + p)
    U, _ = np.linalg.qr(rs.standard_normal((n, p)))
    V, _ = np.linalg.qr(rs.standard_normal((p, p)))
    Y = (U * np.logspace(0, -np.log10(cond), p)) @ V.T             # singular values 1 .. 1/cond
    Q, rdiag = ops.cholqr(torch.from_numpy(Y).to(dev))
    Q, rdiag = Q.cpu().numpy(), rdiag.cpu().numpy()
    R = np.linalg.cholesky(Y.T @ Y).T
    np.testing.assert_allclose(rdiag, np.diag(R), rtol=1e-13 * cond ** 2 + 1e-10)
    # CholeskyQR loses orthogonality like eps cond^2; span(Q) = span(Y) to rounding
    assert np.abs(Q.T @ Q - np.eye(p)).max() <= 1e-15 * cond ** 2 + 1e-12
    Qr, Rr = np.linalg.qr(Y)
    Qr = Qr * np.sign(np.diag(Rr))
    assert np.abs(Q - Qr).max() <= 1e-14 * cond ** 2 + 1e-11


def test_cholqr_zeroes_dependent_columns(dev):
    """A numerically dependent basis column (pivot lost > 13 digits) gives an exactly zero column of Q instead of
    noise (the subspace shrinks); the others stay orthonormal."""
    from ganspace_amd import ops
    rs = np.random.RandomState(3)
    Y = rs.standard_normal((300, 64))
    Y[:, 40] = Y[:, 3] * 2.0 - Y[:, 17]          # exact linear dependence
    Y[:, 63] = Y[:, 62]
    Q, rdiag = ops.cholqr(torch.from_numpy(Y).to(dev))
    Q, rdiag = Q.cpu().numpy(), rdiag.cpu().numpy()
    assert rdiag[40] == 0.0 and rdiag[63] == 0.0 and (np.delete(rdiag, [40, 63]) > 0).all()
    live = np.delete(np.arange(64), [40, 63])
    assert np.abs(Q[:, [40, 63]]).max() == 0.0
    assert np.abs(Q[:, live].T @ Q[:, live] - np.eye(62)).max() < 1e-10


@pytest.mark.parametrize("p,kind", [(128, "dense"), (128, "neardiag"), (128, "clustered"), (112, "dense"),
                                    (96, "dense"), (80, "dense"), (64, "lowrank"), (16, "dense"), (8, "dense")])
def test_jacobi_small_matches_lapack(dev, p, kind):
    from ganspace_amd import ops
    rs = np.random.RandomState(p + len(kind))
    if kind == "dense":
        B = _spd(p, p, 1e4)
    elif kind == "neardiag":
        B = np.diag(np.logspace(0, -3, p)) + 1e-6 * rs.standard_normal((p, p))
        B = (B + B.T) / 2
    elif kind == "clustered":
        Q, _ = np.linalg.qr(rs.standard_normal((p, p)))
        ev = np.repeat(np.logspace(0, -2, p // 4), 4) * (1 + 1e-9 * rs.standard_normal(p))
        B = (Q * ev) @ Q.T
    else:
        A = rs.standard_normal((p // 4, p))
        B = A.T @ A
    theta, U, sweeps, limit = ops.jacobi_small(torch.from_numpy(B).to(dev))
    theta, U = theta.cpu().numpy(), U.cpu().numpy()
    assert not limit and 1 <= sweeps <= 24        # a dense random basis with cond 1e4 needs ~17 cyclic sweeps
    ref = np.sort(np.abs(np.linalg.eigvalsh(B)))[::-1]
    scale = ref[0]
    np.testing.assert_allclose(theta, ref, atol=1e-11 * scale)
    live = theta > 1e-9 * scale
    Ul = U[:, live]
    assert np.abs(Ul.T @ Ul - np.eye(live.sum())).max() < 1e-10       # orthonormal eigenvectors (as columns)
    assert np.abs(B @ Ul - Ul * theta[live]).max() < 1e-10 * scale       # B u = theta u


def _cfg2_like_cov(n_rows=40000, seed=0):
    """Covariance of the random-init mapping network's W vectors (the spectrum BASELINE cfg2 solves)."""
    from oracle import synth
    rs = np.random.RandomState(seed)
    W = (rs.standard_normal((8, 512, 512)) / 0.01).astype(np.float32)
    b = np.zeros((8, 512), np.float32)
    z = rs.standard_normal((n_rows, 512)).astype(np.float32)
    w = synth.mapping_network(z, W, b, dtype=np.float32).astype(np.float64)
    wc = w - w.mean(0)
    return wc.T @ wc


def _check_topk(dev, A, k, expect_converged=True, V0=None, ncheck=None, tol=1e-8):
    from ganspace_amd import ops
    ev, EV = np.linalg.eigh(A)
    ev, EV = ev[::-1], EV[:, ::-1]
    w, V, info = ops.eigh_topk(torch.from_numpy(A).to(dev), k, None if V0 is None else torch.from_numpy(V0).to(dev))
    w, V = w.cpu().numpy(), V.cpu().numpy()
    assert info["converged"] == expect_converged, info
    np.testing.assert_allclose(w, ev[:k], atol=tol * ev[0])
    ncheck = ncheck or k
    res = np.linalg.norm(A @ V[:ncheck].T - V[:ncheck].T * w[:ncheck], axis=0).max() / ev[0]
    assert res < 10 * tol, res
    assert np.abs(V @ V.T - np.eye(k)).max() < 1e-8
    return info


def test_eigh_topk_cfg2_spectrum(dev):
    A = _cfg2_like_cov()
    info = _check_topk(dev, A, 80)
    assert info["subspace"] == 128 and info["products"] <= 20, info        # round 1 needed 26 products


def test_eigh_topk_warm_start(dev):
    """Faithful-mode situation: previous components seed the subspace of a slightly perturbed matrix."""
    A = _cfg2_like_cov()
    ev, EV = np.linalg.eigh(A)
    V0 = EV[:, ::-1][:, :80].T.copy()
    rs = np.random.RandomState(9)
    X = rs.standard_normal((2000, 512)) * 0.3
    A2 = A + X.T @ X
    info = _check_topk(dev, A2, 80, V0=V0)
    assert info["products"] <= 20, info


@pytest.mark.parametrize("name", ["d512_k20", "d512_k80_nb10000"])
def test_eigh_topk_fixture_spectra(dev, name):
    """Steep spectra (lambda_1 / lambda_p ~ 1e6) sitting on a flat noise floor."""
    case = gin.IPCA_CASES[name]
    X = np.concatenate(list(gin.ipca_blocks(case))).astype(np.float64)
    Xc = X - X.mean(0)
    _check_topk(dev, Xc.T @ Xc, case["k"], ncheck=case["ncheck"])


def test_eigh_topk_rank_deficient(dev):
    rs = np.random.RandomState(5)
    A_ = rs.standard_normal((100, 512)) * (1.1 ** -np.arange(100))[:, None]
    X = rs.standard_normal((3000, 100)) @ A_
    _check_topk(dev, X.T @ X, 80, expect_converged=True, ncheck=60, tol=1e-7)


def test_eigh_topk_white_noise_falls_back(dev):
    """No gap after the k-th eigenvalue: the filter cannot converge; the full Jacobi solver must take over and
    still deliver the exact leading pairs."""
    rs = np.random.RandomState(5)
    X = rs.standard_normal((4000, 512))
    _check_topk(dev, X.T @ X, 80, expect_converged=False, ncheck=1)


@pytest.mark.parametrize("M,N,K,ta,tb", [(512, 96, 512, False, False), (96, 96, 512, True, False),
                                         (512, 96, 96, False, False), (2081, 80, 2081, False, False),
                                         (80, 80, 2081, True, False), (80, 512, 80, False, False),
                                         (33, 17, 5, False, True), (1, 1, 1, False, False), (100, 260, 37, True, True),
                                         (2100, 2100, 64, False, True), (130, 131, 4100, True, False),
                                         # tall-skinny products (the small side's T Q shape class): ragged rows / columns / K,
                                         # transposed operands
                                         (2050, 37, 1030, True, True), (2081, 128, 2081, False, False),
                                         (4100, 17, 1100, False, True), (2049, 113, 1025, True, False),
                                         # more tall-times-narrow shapes with whole 16-column tiles, ragged K, every operand layout
                                         (2080, 96, 2080, False, False), (2080, 80, 2080, False, False),
                                         (1024, 16, 520, False, False), (4096, 48, 1000, False, True),
                                         (2048, 64, 2048, True, False), (1056, 32, 515, True, True),
                                         (2080, 128, 2080, False, False), (1024, 112, 777, False, True), (1030, 16, 2081, True, False)])
def test_gemm_f64_mfma_matches_numpy(dev, M, N, K, ta, tb):
    """The f64 matrix-pipe product (`mm64_kernel`, v_mfma_f64_16x16x4_f64) of the solver chains, every operand layout
    (strided views: no transposed copies), ragged shapes, both tile arrangements and the split-K path."""
    from ganspace_amd import ops
    rs = np.random.RandomState(M + 3 * N + 7 * K)
    # asymmetric operands: a transposed C-write or a swapped operand map cannot cancel out
    A = rs.standard_normal((M, K)) * np.linspace(1.0, 2.0, K)[None, :] + np.arange(M)[:, None] * 1e-3
    B = rs.standard_normal((K, N)) + np.arange(N)[None, :] * 1e-2
    At = torch.from_numpy(np.ascontiguousarray(A.T)).to(dev).T if ta else torch.from_numpy(A).to(dev)
    Bt = torch.from_numpy(np.ascontiguousarray(B.T)).to(dev).T if tb else torch.from_numpy(B).to(dev)
    ref = A @ B
    scale = np.abs(A) @ np.abs(B) + 1e-300
    C = ops.gemm_f64(At, Bt).cpu().numpy()
    assert (np.abs(C - ref) / scale).max() < 1e-14
    # alpha / beta form on a padded C
    C0 = rs.standard_normal((M, N + 3))
    Cd = torch.from_numpy(C0).to(dev)
    ops.gemm_f64(At, Bt, alpha=-0.5, beta=2.0, C=Cd[:, :N])
    out = Cd.cpu().numpy()
    assert (np.abs(out[:, :N] - (2.0 * C0[:, :N] - 0.5 * ref)) / (scale + np.abs(C0[:, :N]))).max() < 1e-14
    assert (out[:, N:] == C0[:, N:]).all()
    # three-term (Chebyshev) epilogue
    E1, E2 = rs.standard_normal((M, N)), rs.standard_normal((M, N))
    coef = np.array([0.75, -1.25, 0.5])
    Ce = ops.gemm_f64(At, Bt, coef=torch.from_numpy(coef).to(dev), E1=torch.from_numpy(E1).to(dev),
                      E2=torch.from_numpy(E2).to(dev)).cpu().numpy()
    assert (np.abs(Ce - (0.75 * ref - 1.25 * E1 + 0.5 * E2)) / (scale + 2.0)).max() < 1e-14


@pytest.mark.parametrize("p,kind", [(128, "dense"), (128, "neardiag"), (128, "pca"), (112, "dense"), (96, "dense"),
                                    (96, "pca"), (80, "dense"), (64, "lowrank"), (16, "dense"), (8, "dense"),
                                    (96, "tridiagonal")])
def test_tridiag_eigensolver_matches_lapack(dev, p, kind):
    """`tridiag_eig` (csrc/gs_tridiag.hip: Householder tridiagonalisation, 65-section bisection, twisted-factorisation
    eigenvectors, back-transformation) - the projection step of gs_eigh_topk since round 4 - against LAPACK on the same
    matrix: eigenvalues to rounding, residuals ||B u - theta u|| and orthogonality at the level the well-separated
    spectra of these cases allow (eps ||B|| / gap)."""
    from ganspace_amd import ops
    rs = np.random.RandomState(p + len(kind))
    if kind == "dense":
        B = _spd(p, p, 1e4)
    elif kind == "neardiag":
        B = np.diag(np.logspace(0, -3, p)) + 1e-6 * rs.standard_normal((p, p))
        B = (B + B.T) / 2
    elif kind == "pca":
        # the spectrum the solver sees in cfg2: a decaying covariance with ~1 % spacing in the tail
        Q, _ = np.linalg.qr(rs.standard_normal((p, p)))
        B = (Q * (1.0 / (1.0 + 0.25 * np.arange(p)) ** 1.5)) @ Q.T
    elif kind == "tridiagonal":
        B = np.diag(np.linspace(1.0, 2.0, p)) + np.diag(0.01 * np.ones(p - 1), 1) + np.diag(0.01 * np.ones(p - 1), -1)
    else:
        G = rs.standard_normal((p, p // 2))
        B = G @ G.T + 1e-9 * np.eye(p)
    theta, U, status = ops.eig_tridiag(torch.from_numpy(B).to(dev))
    theta, U = theta.cpu().numpy(), U.cpu().numpy()
    ev = np.linalg.eigvalsh(B)[::-1]
    scale = np.abs(ev).max()
    gaps = np.abs(np.diff(ev)).min() / scale
    if kind == "lowrank":
        assert status == 2            # a (numerically) repeated eigenvalue: reported, the caller uses the Jacobi kernel
        return
    assert status == 0, (status, gaps)
    np.testing.assert_allclose(theta, ev, atol=2e-14 * scale * p)
    res = np.abs(B @ U - U * theta).max() / scale
    assert res < 1e-13 * p, res
    orth = np.abs(U.T @ U - np.eye(p)).max()
    assert orth < max(1e-12, 50 * 2.2e-16 / gaps), (orth, gaps)

"""NumPy restatement of ``fbpca.pca`` for real matrices with ``raw=True`` (TEST INFRASTRUCTURE).

The reference's ``FacebookPCAEstimator.fit`` (``/root/reference/estimators.py:137``) calls
``fbpca.pca(X, k=n_components, n_iter=2, raw=True, l=2 * n_components)``.  fbpca (facebook/fbpca, version 1.0 on
PyPI; ``environment.yml`` of the reference lists it unpinned) is NOT installed in this image and not part of the
reference tree, so this restates its published algorithm - Halko, Martinsson, Tropp, "Finding structure with
randomness" (SIAM Review 53(2), 2011), algorithms 4.4 + 5.1, as implemented in ``fbpca.py:pca`` - branch by branch:

* ``l >= m / 1.25 or l >= n / 1.25``: dense SVD of ``A``;
* ``m >= n``: ``Q = A Omega`` with ``Omega = uniform(-1, 1, (n, l))``, LU-normalised power iterations
  ``Q = A^T Q ; Q = A Q``, last one QR, then ``SVD(Q^T A)``;
* ``m < n``: ``Q = (Omega A)^T`` with ``Omega = uniform(-1, 1, (l, m))``, iterations ``Q = A Q ; Q = A^T Q``,
  then ``SVD(A Q)`` and ``Va = Ra Q^T``.

The test matrix comes from NumPy's GLOBAL legacy stream exactly as in fbpca (``np.random.uniform``), cast to the
dtype of ``A``.  **Parity unpinned**: the reference holds no golden output of this estimator and the package itself is
absent; the restatement is checked against the exact SVD (leading singular values / vectors of matrices with a gap).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import lu, qr, svd


def pca(A: np.ndarray, k: int = 6, raw: bool = True, n_iter: int = 2, l: int | None = None, omega=None):
    """Returns ``(U[:, :k], s[:k], Va[:k, :])`` like ``fbpca.pca``; ``omega`` overrides the random test matrix."""
    if not raw:
        raise NotImplementedError("the reference only uses raw=True")
    if l is None:
        l = k + 2
    m, n = A.shape
    assert 0 < k <= min(m, n) and n_iter >= 0 and l >= k
    if l >= m / 1.25 or l >= n / 1.25:
        U, s, Va = svd(A, full_matrices=False)
        return U[:, :k], s[:k], Va[:k, :]
    if m >= n:
        Om = np.random.uniform(low=-1.0, high=1.0, size=(n, l)) if omega is None else np.asarray(omega)
        Q = A @ Om.astype(A.dtype)
        if n_iter == 0:
            Q, _ = qr(Q, mode="economic")
        else:
            Q, _ = lu(Q, permute_l=True)
        for it in range(n_iter):
            Q = (Q.T @ A).T
            Q, _ = lu(Q, permute_l=True)
            Q = A @ Q
            if it + 1 < n_iter:
                Q, _ = lu(Q, permute_l=True)
            else:
                Q, _ = qr(Q, mode="economic")
        QA = Q.T @ A
        R, s, Va = svd(QA, full_matrices=False)
        U = Q @ R
        return U[:, :k], s[:k], Va[:k, :]
    Om = np.random.uniform(low=-1.0, high=1.0, size=(l, m)) if omega is None else np.asarray(omega)
    Q = (Om.astype(A.dtype) @ A).T
    if n_iter == 0:
        Q, _ = qr(Q, mode="economic")
    else:
        Q, _ = lu(Q, permute_l=True)
    for it in range(n_iter):
        Q = A @ Q
        Q, _ = lu(Q, permute_l=True)
        Q = (Q.T @ A).T
        if it + 1 < n_iter:
            Q, _ = lu(Q, permute_l=True)
        else:
            Q, _ = qr(Q, mode="economic")
    U, s, Ra = svd(A @ Q, full_matrices=False)
    Va = Ra @ Q.T
    return U[:, :k], s[:k], Va[:k, :]


class FacebookPCAEstimatorOracle:
    """Restatement of the reference wrapper (``estimators.py:124-160``) around :func:`pca`."""

    def __init__(self, n_components):
        self.n_components = n_components
        self.n_iter = 2
        self.l = 2 * n_components
        self.batch_support = False

    def get_param_str(self):
        return "fbpca_c{}_it{}_l{}".format(self.n_components, self.n_iter, self.l)

    def fit(self, X):
        _, s, Va = pca(X, k=self.n_components, n_iter=self.n_iter, raw=True, l=self.l)
        self.components_ = np.array(Va)
        self.singular_values_ = s
        self.total_var = X.var(axis=0).sum()
        self.stdev = np.dot(self.components_, X.T).std(axis=1)
        idx = np.argsort(self.stdev)[::-1]
        self.stdev = self.stdev[idx]
        self.components_[:] = self.components_[idx]
        self.mean_ = X.mean(axis=0, keepdims=True)

    def get_components(self):
        return self.components_, self.stdev, self.stdev ** 2 / self.total_var

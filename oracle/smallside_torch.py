"""float64 restatement of the IncrementalPCA recurrence handled from the SMALL side of its stacked matrix,
written with plain ``torch`` matmuls on whatever device the blocks live on (TEST INFRASTRUCTURE).

The CPU oracles of ``oracle/ipca.py`` cost 19 s (d = 32 768) to 83 s (d = 131 072) per 2 000-row block
(SURVEY.md 8d [probe]); this one gives the same answer - it is the algebra of
``IncrementalPCA.partial_fit`` (``sklearn/decomposition/_incremental_pca.py:335-378``, reached through
``/root/reference/estimators.py:68-76``) on the r x r matrix ``M M^T`` of the stack
``M = [diag(S) V ; X - bm ; mc]`` ("small side") - in well under a second per block at
the benchmarked shape (NB = 2000, k = 80, r = 2081), so the parity tests and ``bench.py`` can afford it
where the product's small-side kernels are actually timed.  float64 throughout; the r x r eigenproblem
is LAPACK's (``numpy.linalg.eigh`` on the host).

Checked against ``SklearnRecurrenceOracle`` (the SVD form, the arithmetic the reference executes) in
``tests/test_oracle.py``.  Nothing under ``ganspace_amd/`` imports this module.
"""
from __future__ import annotations

import numpy as np
import torch


class SmallSideTorchOracle:
    """Attribute names follow scikit-learn; arrays are returned as host ``numpy`` float64."""

    def __init__(self, n_components: int):
        self.n_components = int(n_components)
        self.n_samples_seen_ = 0
        self._mean = None        # float64 [d]
        self._m2 = None          # per-feature sum of squared deviations
        self._S = None           # float64 [k]
        self._V = None           # float64 [k, d]

    def partial_fit(self, X):
        X = torch.as_tensor(X)
        Xd = X.to(torch.float64)
        m, d = Xd.shape
        k = self.n_components
        if k > d:
            raise ValueError(f"n_components={k} invalid for n_features={d}")
        if self._V is None and k > m:
            raise ValueError(f"n_components={k} must be less or equal to the batch number of samples {m} "
                             "for the first partial_fit call.")
        bs = Xd.sum(dim=0)
        bm = bs / m
        Xc = Xd - bm
        n0 = self.n_samples_seen_
        n1 = n0 + m
        colsq = (Xc * Xc).sum(dim=0)
        if n0 == 0:
            mean, m2 = bm, colsq
            M = Xc
        else:
            mean = (n0 * self._mean + bs) / n1
            delta = bm - self._mean
            m2 = self._m2 + colsq + delta * delta * (n0 * m / n1)
            mc = np.sqrt(n0 / n1 * m) * (self._mean - bm)
            M = torch.cat([self._S[:, None] * self._V, Xc, mc[None, :]], dim=0)
        T = M @ M.T
        w, U = np.linalg.eigh(T.cpu().numpy())
        order = np.argsort(w)[::-1][:k]
        w = np.maximum(w[order], 0.0)
        Uk = torch.from_numpy(np.ascontiguousarray(U[:, order])).to(M.device)        # [r, k]
        wk = torch.from_numpy(w).to(M.device)
        inv = torch.where(wk > 0, 1.0 / torch.sqrt(torch.clamp(wk, min=1e-300)), torch.zeros_like(wk))
        V = (Uk.T @ M) * inv[:, None]
        # svd_flip(u_based_decision=False): largest-magnitude entry of every row positive (extmath.py:943-951)
        j = torch.argmax(V.abs(), dim=1)
        sgn = torch.sign(V[torch.arange(k, device=V.device), j])
        V = V * sgn[:, None]
        self.n_samples_seen_ = n1
        self._mean, self._m2, self._S, self._V = mean, m2, torch.sqrt(wk), V
        return self

    # -- sklearn attribute surface ----------------------------------------------------------------
    @property
    def components_(self):
        return self._V.cpu().numpy()

    @property
    def singular_values_(self):
        return self._S.cpu().numpy()

    @property
    def mean_(self):
        return self._mean.cpu().numpy()

    @property
    def var_(self):
        return (self._m2 / self.n_samples_seen_).cpu().numpy()

    @property
    def explained_variance_(self):
        return (self._S ** 2 / (self.n_samples_seen_ - 1)).cpu().numpy()

    @property
    def explained_variance_ratio_(self):
        return (self._S ** 2 / self._m2.sum()).cpu().numpy()


def lowrank_plus_noise_blocks(d, n_blocks, rows=2000, latent=128, decay=1.03, noise=0.05, offset=0.3, seed=7,
                              device="cpu"):
    """The synthetic wide-feature workload of SURVEY.md 8d item 5 (``X = G_128 L + noise``): a generator of
    ``[rows, d]`` float32 blocks on ``device``, deterministic in ``seed``."""
    g = torch.Generator(device=device).manual_seed(seed)
    A = torch.randn(latent, d, device=device, generator=g) * (decay ** -torch.arange(latent, device=device,
                                                                                       dtype=torch.float32))[:, None]
    for _ in range(n_blocks):
        X = torch.randn(rows, latent, device=device, generator=g) @ A
        X += noise * torch.randn(rows, d, device=device, generator=g)
        X += offset
        yield X

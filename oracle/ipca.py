"""Oracle restatements of the PCA arithmetic on the hot path (TEST INFRASTRUCTURE).

Three independent CPU statements of "what the right answer is":

``SklearnRecurrenceOracle``
    The stacked-matrix SVD recurrence that the reference executes through
    ``estimators.IPCAEstimator.fit_partial`` (``/root/reference/estimators.py:68-76``)
    -> ``sklearn.decomposition.IncrementalPCA.partial_fit``
    (``sklearn/decomposition/_incremental_pca.py:257-379``, scikit-learn 1.7.2):
    running mean / variance (``sklearn/utils/extmath.py:1064-1187``), centring,
    ``vstack([S*V ; Xc ; mean_correction])``, thin SVD (LAPACK gesdd),
    ``svd_flip(u_based_decision=False)`` (``extmath.py:895-953``), truncation.
    The dtype behaviour is restated too: block 1 runs in the input dtype
    (float32 on the reference path), later blocks are promoted to float64 by
    the float64 mean-correction row (SURVEY.md §3.4 [probe]).

``GramRecurrenceOracle``
    The same recurrence expressed on the d x d matrix ``M^T M`` of that stack
     in float64: this is the algebra the HIP library executes
    ("ipca-faithful" mode) and is what the device results are compared to at
    tight tolerance.

``exact_pca``
    One global centred scatter + one symmetric eigendecomposition ("exact"
    mode, the north_star's Gram/all-reduce/eigensolve design).

None of this is imported by the product package.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg


def flip_rows_largest_abs_positive(vt: np.ndarray) -> np.ndarray:
    """Sign convention of ``svd_flip(u, v, u_based_decision=False)``.

    Follows ``sklearn/utils/extmath.py:943-951``: for every row of ``vt`` the
    entry of largest magnitude (first one on ties, ``argmax`` semantics) is
    made positive.  Returns the per-row signs (+1/-1, 0 for an all-zero row).
    """
    j = np.argmax(np.abs(vt), axis=1)
    return np.sign(vt[np.arange(vt.shape[0]), j])


def _column_stats_update(X, last_mean, last_var, last_n):
    """Chan/Golub/LeVeque update of per-column mean and (biased) variance.

    Restates ``_incremental_mean_and_var`` (``extmath.py:1120-1187``) for the
    unweighted, NaN-free case; all accumulators are float64 as in
    ``_safe_accumulator_op``.
    """
    m = X.shape[0]
    new_sum = X.sum(axis=0, dtype=np.float64)
    n_tot = last_n + m
    last_sum = last_mean * last_n
    mean = (last_sum + new_sum) / n_tot
    t = new_sum / m
    dev = X - t                      # float32 stays float32 - t is float64 -> float64
    corr = dev.sum(axis=0, dtype=np.float64)
    m2_new = (dev * dev).sum(axis=0, dtype=np.float64) - corr ** 2 / m
    if last_n == 0:
        m2 = m2_new
    else:
        ratio = last_n / m
        m2 = (last_var * last_n + m2_new
              + ratio / n_tot * (last_sum / ratio - new_sum) ** 2)
    return mean, m2 / n_tot, n_tot


class SklearnRecurrenceOracle:
    """SVD-form restatement of ``IncrementalPCA(k, whiten=False).partial_fit``.

    Attribute names match scikit-learn so a test can diff them directly.
    """

    def __init__(self, n_components: int):
        self.n_components = int(n_components)
        self.n_samples_seen_ = 0
        self.mean_ = 0.0
        self.var_ = 0.0
        self.components_ = None
        self.singular_values_ = None

    def partial_fit(self, X: np.ndarray):
        X = np.array(X, copy=True)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        m, d = X.shape
        k = self.n_components
        # error behaviour of _incremental_pca.py:300-314
        if k > d:
            raise ValueError(f"n_components={k} invalid for n_features={d}")
        if self.components_ is None and k > m:
            raise ValueError(f"n_components={k} must be less or equal to the "
                             f"batch number of samples {m} for the first partial_fit call.")

        mean, var, n_tot = _column_stats_update(X, self.mean_, self.var_, self.n_samples_seen_)
        if self.n_samples_seen_ == 0:
            X -= mean                               # in place: keeps the input dtype
            stack = X
        else:
            batch_mean = X.mean(axis=0)             # input dtype, as np.mean does
            X -= batch_mean
            corr = np.sqrt((self.n_samples_seen_ / n_tot) * m) * (self.mean_ - batch_mean)
            stack = np.vstack((self.singular_values_[:, None] * self.components_, X, corr[None, :]))

        _, S, Vt = scipy.linalg.svd(stack, full_matrices=False, check_finite=False)
        Vt = Vt * flip_rows_largest_abs_positive(Vt)[:, None]
        ev = S ** 2 / (n_tot - 1)
        evr = S ** 2 / np.sum(var * n_tot)

        self.n_samples_seen_ = n_tot
        self.components_ = Vt[:k]
        self.singular_values_ = S[:k]
        self.mean_, self.var_ = mean, var
        self.explained_variance_ = ev[:k]
        self.explained_variance_ratio_ = evr[:k]
        self.noise_variance_ = ev[k:].mean() if k not in (m, d) else 0.0
        return self


class GramRecurrenceOracle:
    """float64 d x d Gram form of the same recurrence.

    ``stack^T stack = V^T diag(S^2) V + Xc^T Xc + mc mc^T`` so the right
    singular vectors / squared singular values of the stack are the
    eigenpairs of that d x d matrix.  This is exactly what
    ``ganspace_amd/csrc`` computes on the device.
    """

    def __init__(self, n_components: int):
        self.n_components = int(n_components)
        self.n_samples_seen_ = 0
        self.mean_ = None
        self.m2_ = None                 # per-column sum of squared deviations
        self.components_ = None
        self.singular_values_ = None

    def partial_fit(self, X: np.ndarray):
        X = np.asarray(X, dtype=np.float64)
        m, d = X.shape
        k = self.n_components
        if k > d:
            raise ValueError(f"n_components={k} invalid for n_features={d}")
        if self.components_ is None and k > m:
            raise ValueError(f"n_components={k} must be less or equal to the "
                             f"batch number of samples {m} for the first partial_fit call.")
        bs = X.sum(axis=0)
        bm = bs / m
        Xc = X - bm
        Gc = Xc.T @ Xc
        n0 = self.n_samples_seen_
        n1 = n0 + m
        if n0 == 0:
            mean = bm
            m2 = np.diag(Gc).copy()
        else:
            mean = (n0 * self.mean_ + bs) / n1
            delta = bm - self.mean_
            m2 = self.m2_ + np.diag(Gc) + delta ** 2 * (n0 * m / n1)
            mc = np.sqrt(n0 / n1 * m) * (self.mean_ - bm)
            SV = self.singular_values_[:, None] * self.components_
            Gc = Gc + SV.T @ SV + np.outer(mc, mc)
        w, U = np.linalg.eigh(Gc)
        order = np.argsort(w)[::-1][:k]
        w = np.maximum(w[order], 0.0)
        Vt = U[:, order].T
        Vt = Vt * flip_rows_largest_abs_positive(Vt)[:, None]

        self.n_samples_seen_ = n1
        self.mean_, self.m2_ = mean, m2
        self.var_ = m2 / n1
        self.components_ = Vt
        self.singular_values_ = np.sqrt(w)
        self.explained_variance_ = w / (n1 - 1)
        self.explained_variance_ratio_ = w / np.sum(m2)
        return self


def exact_pca(blocks, n_components: int):
    """Exact covariance PCA of the concatenation of ``blocks`` (float64).

    Per-block centred scatter + pairwise (Chan) merge, one ``eigh`` at the end;
    same sign convention and the same derived quantities as sklearn reports.
    Returns a dict with sklearn's attribute names.
    """
    n = 0
    mean = None
    C = None
    for X in blocks:
        X = np.asarray(X, dtype=np.float64)
        m = X.shape[0]
        bm = X.mean(axis=0)
        Xc = X - bm
        Cb = Xc.T @ Xc
        if n == 0:
            n, mean, C = m, bm, Cb
        else:
            delta = bm - mean
            C = C + Cb + np.outer(delta, delta) * (n * m / (n + m))
            mean = (n * mean + m * bm) / (n + m)
            n += m
    w, U = np.linalg.eigh(C)
    order = np.argsort(w)[::-1][:n_components]
    w = np.maximum(w[order], 0.0)
    Vt = U[:, order].T
    Vt = Vt * flip_rows_largest_abs_positive(Vt)[:, None]
    return dict(components_=Vt, singular_values_=np.sqrt(w), mean_=mean,
                var_=np.diag(C) / n, n_samples_seen_=n,
                explained_variance_=w / (n - 1),
                explained_variance_ratio_=w / np.trace(C))


class IPCAEstimatorOracle:
    """Restatement of the reference wrapper ``IPCAEstimator``
    (``/root/reference/estimators.py:55-81``) around an oracle transformer.

    ``kind`` selects ``'svd'`` (SklearnRecurrenceOracle) or ``'gram'``
    (GramRecurrenceOracle).
    """

    def __init__(self, n_components: int, kind: str = "svd"):
        self.n_components = n_components
        self.whiten = False
        self.batch_support = True
        cls = {"svd": SklearnRecurrenceOracle, "gram": GramRecurrenceOracle}[kind]
        self.transformer = cls(n_components)

    def get_param_str(self):
        return "ipca_c{}{}".format(self.n_components, "_w" if self.whiten else "")

    def fit_partial(self, X):
        try:
            self.transformer.partial_fit(X)
            self.transformer.n_samples_seen_ = np.int64(self.transformer.n_samples_seen_)
            return True
        except ValueError as e:           # estimators.py:74-76: swallow, tell the loop to stop
            print("\nIPCA error:", e)
            return False

    def get_components(self):
        t = self.transformer
        return t.components_, np.sqrt(t.explained_variance_), t.explained_variance_ratio_


def signed_cosines(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Row-wise signed cosine between two component matrices ``[k, d]``."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    num = np.sum(A * B, axis=1)
    return num / (np.linalg.norm(A, axis=1) * np.linalg.norm(B, axis=1))


# ---- multi-rank merge of the sklearn-faithful recurrence (new design) -------------------------------
def pack_lowrank_state(t) -> np.ndarray:
    """Low-rank state of a fitted recurrence oracle / sklearn ``IncrementalPCA`` ``t`` in the layout of
    ``gs_ipca_lowrank_export``: float64 ``[n | mean(d) | m2(d) | lam(k) | V(k x d)]`` with ``m2 = var_ * n`` and
    ``lam = singular_values_ ** 2``."""
    n = float(t.n_samples_seen_)
    mean = np.asarray(t.mean_, dtype=np.float64)
    V = np.asarray(t.components_, dtype=np.float64)
    return np.concatenate([[n], mean, np.asarray(t.var_, dtype=np.float64) * n,
                           np.asarray(t.singular_values_, dtype=np.float64) ** 2, V.ravel()])


def merge_lowrank_states(states, k: int, d: int):
    """ONE more step of ``IncrementalPCA.partial_fit`` (``_incremental_pca.py:335-378``) with every rank's state as a
    pre-compressed batch: the vstack of :347-362 becomes ``[sqrt(lam_r) V_r ; sqrt(n_r) (mean_r - mean)]`` over all
    ranks - the scatter about the global mean decomposes exactly into the per-rank scatters (here: their rank-k
    truncations ``V_r^T diag(lam_r) V_r``) plus ``n_r (mean_r - mean)(mean_r - mean)^T`` (Chan et al.; P = 2 gives
    sklearn's own correction row ``sqrt(n0 m / n1) (mean0 - mean1)`` split over the two means) - then thin SVD,
    ``svd_flip(u_based_decision=False)``, truncation.  Returns a dict with sklearn's attribute names."""
    states = [np.asarray(s, dtype=np.float64) for s in states]
    ns = np.array([s[0] for s in states])
    n = ns.sum()
    means = np.stack([s[1:1 + d] for s in states])
    mean = (ns[:, None] * means).sum(0) / n
    m2 = sum(s[1 + d:1 + 2 * d] + nr * (mu - mean) ** 2 for s, nr, mu in zip(states, ns, means) if nr > 0)
    rows = []
    for s, nr, mu in zip(states, ns, means):
        if nr <= 0:
            continue
        lam = np.maximum(s[1 + 2 * d:1 + 2 * d + k], 0.0)
        V = s[1 + 2 * d + k:].reshape(k, d)
        rows.append(np.sqrt(lam)[:, None] * V)
        rows.append(np.sqrt(nr) * (mu - mean)[None, :])
    stack = np.vstack(rows)
    _, S, Vt = scipy.linalg.svd(stack, full_matrices=False, check_finite=False)
    Vt = Vt * flip_rows_largest_abs_positive(Vt)[:, None]
    return dict(components_=Vt[:k], singular_values_=S[:k], mean_=mean, var_=m2 / n, n_samples_seen_=int(n),
                explained_variance_=S[:k] ** 2 / (n - 1), explained_variance_ratio_=S[:k] ** 2 / m2.sum())

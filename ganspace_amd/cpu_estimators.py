"""CPU pass-through for the two reference estimators that are neither PCA nor batchable.

``ICAEstimator`` (reference estimators.py:18-52) and ``SPCAEstimator`` (:165-204) are whole-matrix
scikit-learn fits (FastICA / SparsePCA) that SURVEY.md §2 keeps on the CPU: they are not on the
"sample -> partial_forward -> incremental PCA" hot path and have no Gram / eigensolver structure to
accelerate.  They live in this separate module so that every estimator name the reference accepts keeps
working (``visualize.py --est=ica``), while nothing on the accelerated path (``'ipca'``, ``'ipca-exact'``,
``'pca'``, ``'fbpca'``) ever imports scikit-learn.  The arithmetic here IS the reference's: the same
scikit-learn classes with the same arguments.
"""
from __future__ import annotations

import numpy as np


def _host(X):
    try:
        import torch
        if torch.is_tensor(X):
            return X.detach().cpu().numpy()
    except ImportError:
        pass
    return np.asarray(X)


class ICAEstimator:
    def __init__(self, n_components):
        from sklearn.decomposition import FastICA
        self.n_components = n_components
        self.maxiter = 10000
        self.whiten = True
        # the reference passes whiten=True (estimators.py:26), which the scikit-learn of its era read as
        # "arbitrary-variance"; current releases only accept the named variants.  The difference is a per-component
        # scale, removed by the row normalisation in fit() below.
        self.transformer = FastICA(n_components, random_state=0, whiten="unit-variance", max_iter=self.maxiter)
        self.batch_support = False
        self.stdev = np.zeros((n_components,))
        self.total_var = 0.0

    def get_param_str(self):
        return "ica_c{}{}".format(self.n_components, "_w" if self.whiten else "")

    def fit(self, X):
        X = _host(X)
        self.transformer.fit(X)
        if self.transformer.n_iter_ >= self.maxiter:
            raise RuntimeError(f"FastICA did not converge (N={X.shape[0]}, it={self.maxiter})")
        comp = self.transformer.components_
        comp /= np.sqrt(np.sum(comp ** 2, axis=-1, keepdims=True))
        self.total_var = X.var(axis=0).sum()
        self.stdev = np.dot(comp, X.T).std(axis=1)
        order = np.argsort(self.stdev)[::-1]
        self.stdev = self.stdev[order]
        comp[:] = comp[order]

    def get_components(self):
        return self.transformer.components_, self.stdev, self.stdev ** 2 / self.total_var


class SPCAEstimator:
    def __init__(self, n_components, alpha=10.0):
        from sklearn.decomposition import SparsePCA
        self.n_components = n_components
        self.whiten = False
        self.alpha = alpha
        # the reference passes normalize_components=True, which current scikit-learn removed (components are
        # always normalised now)
        self.transformer = SparsePCA(n_components, alpha=alpha, ridge_alpha=0.01, max_iter=100, random_state=0,
                                     n_jobs=-1)
        self.batch_support = False
        self.stdev = np.zeros((n_components,))
        self.total_var = 0.0

    def get_param_str(self):
        return "spca_c{}_a{}{}".format(self.n_components, self.alpha, "_w" if self.whiten else "")

    def fit(self, X):
        X = _host(X)
        self.transformer.fit(X)
        self.total_var = X.var(axis=0).sum()
        self.stdev = self.transformer.transform(X).std(axis=0)
        order = np.argsort(self.stdev)[::-1]
        self.stdev = self.stdev[order]
        self.transformer.components_[:] = self.transformer.components_[order]

    def get_components(self):
        return self.transformer.components_, self.stdev, self.stdev ** 2 / self.total_var


def make(name, n_components, alpha):
    if name == "ica":
        return ICAEstimator(n_components)
    if name == "spca":
        return SPCAEstimator(n_components, alpha)
    raise RuntimeError("Unknown estimator")

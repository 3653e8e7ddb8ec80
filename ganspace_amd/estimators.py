"""Estimator protocol of the reference, backed by the gfx950 library.

Mirrors ``/root/reference/estimators.py``: ``get_estimator(name, n_components, alpha)``
(:206-218) returns an object with ``batch_support``, ``fit``, ``fit_partial``,
``get_components``, ``get_param_str`` and a ``transformer`` attribute exposing sklearn's
attribute names (``mean_``, ``components_``, ``explained_variance_`` ...), which is all
``decomposition.compute`` (decomposition.py:192-293) touches.

``'ipca'``        sklearn-faithful recurrence on the device (all k components and signs
                  match ``IncrementalPCA``); cache key ``ipca_c{k}`` as in the reference.
                  Gram-side (d x d) for feat_dim <= 8192, small-side (r x r, r = k+rows+1)
                  beyond, chosen automatically.
``'ipca-exact'``  one global Gram + one eigensolve (the multi-GPU all-reduce design);
                  leading components match to ~1e-6 cosine, trailing ones are the *exact*
                  PCA instead of IPCA's truncated approximation; cache key ``ipca-exact_c{k}``.

There is no CPU fallback: constructing an estimator loads the HIP library and fails
loudly if it is missing.  The non-batch estimators of the reference (pca / fbpca / ica /
spca, estimators.py:18-52,84-204) are whole-matrix CPU fits outside this hot path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


class _DeviceIncrementalPCA:
    """The ``transformer`` object: sklearn ``IncrementalPCA`` attribute surface.

    Fitted attributes are produced by ``gs_ipca_finalize`` on first access after an
    update and cached until the next one.
    """

    GRAM_SIDE_MAX_FEATURES = 8192

    def __init__(self, n_components: int, mode: int, device=None, precision: str = "f32"):
        self._precision = _lib.PRECISIONS[precision]
        self.n_components = int(n_components)
        self.whiten = False
        self.batch_size = max(100, 2 * self.n_components)     # estimators.py:59
        self._mode = mode
        self._device = device
        self._h = None
        self._d = None
        self._cache = None
        self._lib = _lib.load()

    # -- lifetime ---------------------------------------------------------------------
    def _ensure(self, d: int):
        if self._h is not None:
            if d != self._d:
                raise ValueError(f"X has {d} features, but IncrementalPCA is expecting {self._d} features as input.")
            return
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("ganspace_amd needs a HIP device (torch.cuda.is_available() is False)")
        if self._device is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        if self._mode == _lib.GS_MODE_FAITHFUL and d > self.GRAM_SIDE_MAX_FEATURES:
            # feat_dim >> block rows (gen_z d = 32 768, conv features d = 131 072): same recurrence,
            # handled from the small side of the stacked matrix
            self._mode = _lib.GS_MODE_SMALLSIDE
        h = C.c_void_p()
        _lib.check(self._lib.gs_ipca_create(d, self.n_components, self._mode, self._precision,
                                            self._device.index or 0, C.byref(h)))
        self._h, self._d = h, d

    def close(self):
        if self._h is not None:
            self._lib.gs_ipca_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- fitting ------------------------------------------------------------------------
    def _as_device_rows(self, X):
        torch = _torch()
        if isinstance(X, np.ndarray):
            if X.ndim != 2:
                raise ValueError(f"Expected 2D array, got {X.ndim}D array instead")
            X = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32))
        if not torch.is_tensor(X) or X.dim() != 2:
            raise ValueError("Expected a 2D float32 array / tensor")
        dev = self._device or torch.device("cuda", torch.cuda.current_device())
        X = X.to(device=dev, dtype=torch.float32)
        if X.stride(1) != 1:
            X = X.contiguous()
        return X

    def partial_fit(self, X, y=None, check_input=True):
        """One block; ``X`` may be a host ndarray or a device tensor ``[m, d]`` float32."""
        if self.n_components > np.shape(X)[1]:
            raise ValueError(
                f"n_components={self.n_components} invalid for n_features={np.shape(X)[1]}, need more rows "
                "than columns for IncrementalPCA processing")
        Xd = self._as_device_rows(X)
        self._ensure(Xd.shape[1])
        rc = self._lib.gs_ipca_update(self._h, C.c_void_p(Xd.data_ptr()), Xd.shape[0], Xd.stride(0),
                                      _lib.current_stream_ptr())
        if rc == _lib.GS_EINVAL:
            raise ValueError(self._lib.gs_last_error().decode())
        _lib.check(rc)
        # a temporary device copy of a host block is released to torch's caching allocator here;
        # reuse is stream-ordered behind the kernels just enqueued on the current stream
        self._cache = None
        return self

    def fit(self, X, y=None):
        """sklearn ``IncrementalPCA.fit``: batches of ``batch_size`` (last one merged if < k)."""
        if self._h is not None:
            _lib.check(self._lib.gs_ipca_reset(self._h))
        n = np.shape(X)[0]
        bs, k = self.batch_size, self.n_components
        start = 0
        while start < n:
            end = min(start + bs, n)
            if n - end < k:       # sklearn gen_batches(min_batch_size=n_components)
                end = n
            self.partial_fit(X[start:end])
            start = end
        return self

    # -- results -------------------------------------------------------------------------
    def _results(self):
        if self._cache is None:
            if self._h is None:
                raise AttributeError("This IncrementalPCA instance is not fitted yet")
            k, d = self.n_components, self._d
            comp = np.empty((k, d), np.float32)
            sv, ev, evr = (np.empty(k, np.float64) for _ in range(3))
            mean, var = np.empty(d, np.float64), np.empty(d, np.float64)
            n = C.c_int64(0)
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            _lib.check(self._lib.gs_ipca_finalize(self._h, p(comp), p(sv), p(mean), p(var), p(ev), p(evr),
                                                  C.cast(C.byref(n), C.c_void_p), _lib.current_stream_ptr()))
            self._cache = dict(components_=comp, singular_values_=sv, mean_=mean, var_=var,
                               explained_variance_=ev, explained_variance_ratio_=evr,
                               n_samples_seen_=np.int64(n.value))
        return self._cache

    def __getattr__(self, name):
        if name in ("components_", "singular_values_", "mean_", "var_", "explained_variance_",
                    "explained_variance_ratio_", "n_samples_seen_"):
            return self._results()[name]
        raise AttributeError(name)

    @property
    def n_components_(self):
        return self.n_components

    def transform(self, X):
        """``(X - mean_) @ components_.T`` on the device; returns a host ndarray ``[m, k]``."""
        torch = _torch()
        self._results()
        Xd = self._as_device_rows(X)
        comp, mean = C.c_void_p(), C.c_void_p()
        _lib.check(self._lib.gs_ipca_components_device(self._h, C.byref(comp), C.byref(mean)))
        k, d = self.n_components, self._d
        if d % 4 != 0:
            raise NotImplementedError("transform() needs n_features to be a multiple of 4")
        Xd = Xd.contiguous()
        c = torch.from_numpy(self._results()["components_"]).to(Xd.device)
        bias = -(c.double() @ torch.from_numpy(self._results()["mean_"]).to(Xd.device)).float().contiguous()
        out = torch.empty((Xd.shape[0], k), dtype=torch.float32, device=Xd.device)
        _lib.check(self._lib.gs_linear_forward(C.c_void_p(Xd.data_ptr()), comp, C.c_void_p(bias.data_ptr()),
                                               C.c_void_p(out.data_ptr()), Xd.shape[0], d, k,
                                               _lib.current_stream_ptr()))
        return out.cpu().numpy()

    def inverse_transform(self, Y):
        r = self._results()
        return np.asarray(Y, dtype=np.float64) @ r["components_"].astype(np.float64) + r["mean_"]

    # -- multi-GPU / resume: sufficient statistics (EXACT mode) ---------------------------
    def export_state(self):
        """float64 device tensor ``[1 + d + d*d]`` = (n, mean, centred scatter)."""
        torch = _torch()
        if self._h is None:
            raise RuntimeError("nothing fitted yet")
        nb = self._lib.gs_ipca_state_nbytes(self._h)
        st = torch.empty(nb // 8, dtype=torch.float64, device=self._device)
        _lib.check(self._lib.gs_ipca_state_export(self._h, C.c_void_p(st.data_ptr()), _lib.current_stream_ptr()))
        return st

    def import_state(self, state, d=None):
        torch = _torch()
        if self._h is None:
            if d is None:
                d = int(round((-1 + (1 + 4 * (state.numel() - 1)) ** 0.5) / 2))
            self._ensure(d)
        st = state.to(device=self._device, dtype=torch.float64).contiguous()
        _lib.check(self._lib.gs_ipca_state_import(self._h, C.c_void_p(st.data_ptr()), _lib.current_stream_ptr()))
        torch.cuda.current_stream().synchronize()
        self._cache = None
        return self


class IPCAEstimator:
    """Drop-in for the reference ``IPCAEstimator`` (estimators.py:55-81)."""

    def __init__(self, n_components, mode="faithful", device=None, precision="f32"):
        self.n_components = n_components
        self.whiten = False
        self.mode = mode
        m = {"faithful": _lib.GS_MODE_FAITHFUL, "exact": _lib.GS_MODE_EXACT,
             "smallside": _lib.GS_MODE_SMALLSIDE}[mode]
        self.transformer = _DeviceIncrementalPCA(n_components, m, device, precision)
        self.batch_support = True

    def get_param_str(self):
        tag = "ipca-exact" if self.mode == "exact" else "ipca"
        return "{}_c{}{}".format(tag, self.n_components, "_w" if self.whiten else "")

    def fit(self, X):
        self.transformer.fit(X)

    def fit_partial(self, X):
        try:
            self.transformer.partial_fit(X)
            return True
        except ValueError as e:          # estimators.py:74-76
            print(f"\nIPCA error:", e)
            return False

    def get_components(self):
        t = self.transformer
        stdev = np.sqrt(t.explained_variance_)
        return t.components_, stdev, t.explained_variance_ratio_


_OUT_OF_SCOPE = ("pca", "fbpca", "ica", "spca")


def get_estimator(name, n_components, alpha=1.0):
    """Factory with the reference's names and error behaviour (estimators.py:206-218)."""
    if name == "ipca":
        return IPCAEstimator(n_components, "faithful")
    if name == "ipca-exact":
        return IPCAEstimator(n_components, "exact")
    if name in _OUT_OF_SCOPE:
        raise NotImplementedError(
            f"estimator '{name}' is a whole-matrix CPU fit of the reference and is outside the "
            "MI355X batch hot path; use 'ipca' or 'ipca-exact'")
    raise RuntimeError("Unknown estimator")

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

``'pca'``         the reference's whole-matrix ``PCAEstimator`` (estimators.py:84-118) on the device:
                  the same Gram kernel over all rows, one eigensolve; cache key ``pca-full_c{k}``.
``'fbpca'``       the reference's ``FacebookPCAEstimator`` (estimators.py:124-160): the randomized range finder itself
                  (``fbpca.pca(X, k, n_iter=2, raw=True, l=2k)`` of the *uncentred* matrix) as ``gs_randomized_pca`` -
                  2 (n_iter + 1) passes over X on the f32 MFMA, float64 CholeskyQR of the bases, Rayleigh-Ritz from the
                  l x l side, test matrix from NumPy's global stream as fbpca draws it; cache key
                  ``fbpca_c{k}_it2_l{2k}``.  (fbpca is not installed here: parity is pinned to a restatement of its
                  published algorithm, ``oracle/fbpca_port.py``, not to the package.)
``'ica'`` / ``'spca'``  are not batch estimators and not PCA; ``get_estimator`` hands them to the CPU
                  pass-through in ``ganspace_amd/cpu_estimators.py`` (scikit-learn, the reference's own
                  arithmetic) so that every name the reference accepts still works.

There is no CPU fallback on the hot path: constructing any of the four PCA estimators loads the HIP
library and fails loudly if it is missing.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

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
        self._resident_refs = []        # tensors handed over with resident=True, released once their rows are contracted
        self._n_host = 0                # samples fed through this object (None once a state was imported)
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

    def partial_fit(self, X, y=None, check_input=True, resident=False):
        """One block; ``X`` may be a host ndarray or a device tensor ``[m, d]`` float32.

        ``resident=True`` (device tensors only) promises that the rows stay valid and unchanged until the results are
        read - slices of a resident latent array, for instance.  The exact mode then merges contiguous blocks into
        long Gram launches (``gs_ipca_update_resident``); a reference to the tensor is kept until then."""
        if self.n_components > np.shape(X)[1]:
            raise ValueError(
                f"n_components={self.n_components} invalid for n_features={np.shape(X)[1]}, need more rows "
                "than columns for IncrementalPCA processing")
        Xd = self._as_device_rows(X)
        self._ensure(Xd.shape[1])
        torch = _torch()
        if resident and torch.is_tensor(X) and Xd.data_ptr() == X.data_ptr():
            self._resident_refs.append(Xd)       # keeps the storage alive until the next _results()
            rc = self._lib.gs_ipca_update_resident(self._h, C.c_void_p(Xd.data_ptr()), Xd.shape[0], Xd.stride(0),
                                                   _lib.current_stream_ptr())
        else:
            rc = self._lib.gs_ipca_update(self._h, C.c_void_p(Xd.data_ptr()), Xd.shape[0], Xd.stride(0),
                                          _lib.current_stream_ptr())
        if rc == _lib.GS_EINVAL:
            raise ValueError(self._lib.gs_last_error().decode())
        _lib.check(rc)
        if self._n_host is not None:
            self._n_host += int(Xd.shape[0])
        # a temporary device copy of a host block is released to torch's caching allocator here;
        # reuse is stream-ordered behind the kernels just enqueued on the current stream
        self._cache = None
        return self

    def fit(self, X, y=None):
        """sklearn ``IncrementalPCA.fit``: batches of ``batch_size`` (last one merged if < k)."""
        if self._h is not None:
            _lib.check(self._lib.gs_ipca_reset(self._h))
            self._n_host = 0
            self._resident_refs.clear()
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
            self._resident_refs.clear()       # finalize contracted every pending row and synchronised
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
        r = self._results()                # finalizes once (the faithful mode defers its diagonalisation until here)
        Xd = self._as_device_rows(X)
        k, d = self.n_components, self._d
        if Xd.shape[1] != d:
            raise ValueError(f"X has {Xd.shape[1]} features, but IncrementalPCA is expecting {d} features as input.")
        # sklearn's order (``X - mean_`` first, then ``@ components_.T``): the rows are centred while gs_project_rows stages
        # its tiles, so data that sits on a mean many times its spread does not cancel in float32 (X C^T - mean C^T would)
        from . import ops
        if d % 4 == 0:
            if Xd.stride(0) % 4 or Xd.data_ptr() % 16:
                Xd = Xd.contiguous()
            # the components stay where gs_ipca_finalize left them (device, float32 [k, d] + the float32 mean)
            comp, mean = C.c_void_p(), C.c_void_p()
            _lib.check(self._lib.gs_ipca_components_device(self._h, C.byref(comp), C.byref(mean)))
            out = ops.project_rows_ptr(Xd, comp, k, mean)
        else:
            # the kernel reads 16-byte pieces of a row: zero-pad the features (a zero column of X, of the mean and of every
            # component changes nothing - the same device _WholeMatrixPCA.fit uses); padded operands cached with the results
            dp = d + (-d) % 4
            if "_comp_dev" not in r:
                comp_p = np.zeros((k, dp), np.float32)
                comp_p[:, :d] = r["components_"]
                mean_p = np.zeros(dp, np.float32)
                mean_p[:d] = r["mean_"]
                r["_comp_dev"] = torch.from_numpy(comp_p).to(Xd.device)
                r["_mean_dev"] = torch.from_numpy(mean_p).to(Xd.device)
            Xp = torch.zeros((Xd.shape[0], dp), dtype=torch.float32, device=Xd.device)
            Xp[:, :d] = Xd
            out = ops.project_rows(Xp, r["_comp_dev"], shift=r["_mean_dev"])
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
        # (the import reads the sample count on the host and has synchronised the stream: `st` may be released)
        self._n_host = None
        self._cache = None
        return self


    # -- multi-GPU / resume: low-rank state (FAITHFUL / SMALLSIDE modes) ---------------------------------
    def lowrank_len(self):
        """Length (float64 elements) of the low-rank state ``[n | mean(d) | m2(d) | lam(k) | V(k x d)]``."""
        return 1 + 2 * self._d + self.n_components * (1 + self._d)

    def export_lowrank(self):
        """What sklearn's IncrementalPCA carries between two ``partial_fit`` calls, as one float64 device tensor."""
        torch = _torch()
        if self._h is None:
            raise RuntimeError("nothing fitted yet")
        st = torch.empty(self._lib.gs_ipca_lowrank_nbytes(self._h) // 8, dtype=torch.float64, device=self._device)
        _lib.check(self._lib.gs_ipca_lowrank_export(self._h, C.c_void_p(st.data_ptr()), _lib.current_stream_ptr()))
        self._cache = None
        return st

    def merge_lowrank(self, states):
        """Replace this estimator's state by the merge of ``states`` (``[P, lowrank_len]`` float64, e.g. the
        all-gathered states of P ranks): one more step of the recurrence with every state as a pre-compressed batch
        (``gs_ipca_lowrank_merge``)."""
        torch = _torch()
        st = states.to(device=self._device, dtype=torch.float64).contiguous()
        if st.dim() != 2 or st.shape[1] != self.lowrank_len():
            raise ValueError(f"expected [P, {self.lowrank_len()}] states, got {tuple(st.shape)}")
        _lib.check(self._lib.gs_ipca_lowrank_merge(self._h, C.c_void_p(st.data_ptr()), int(st.shape[0]),
                                                   _lib.current_stream_ptr()))
        torch.cuda.current_stream().synchronize()        # `st` may be released
        self._n_host = None
        self._cache = None
        self._resident_refs.clear()
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

    def fit_partial(self, X, resident=False):
        """One block (reference estimators.py:68-76).  Contract for device tensors: the rows are READ ONLY and are read in
        stream order before this call's work on the current stream ends - ``decomposition._fit_blocks`` relies on it when it
        hands over slices of the hooked activation or of the resident latent array instead of a copy.  ``resident=True``
        (rows that stay valid and unchanged until the results are read; the exact mode may then defer and merge their
        contraction) must therefore never be combined with memory the caller is about to reuse."""
        try:
            self.transformer.partial_fit(X, resident=resident)
            return True
        except ValueError as e:          # estimators.py:74-76
            print(f"\nIPCA error:", e)
            return False

    def get_components(self):
        t = self.transformer
        stdev = np.sqrt(t.explained_variance_)
        return t.components_, stdev, t.explained_variance_ratio_


def _column_moments(lib, Xd):
    """``(sum [d], sumsq [d])`` float64 host arrays of the rows of ``Xd`` - one pass of ``gs_column_moments``."""
    torch = _torch()
    n, d = Xd.shape
    acc = torch.zeros((2, d), dtype=torch.float64, device=Xd.device)
    _lib.check(lib.gs_column_moments(C.c_void_p(Xd.data_ptr()), n, Xd.stride(0), d, C.c_void_p(0),
                                     C.c_void_p(acc[0].data_ptr()), C.c_void_p(acc[1].data_ptr()),
                                     _lib.current_stream_ptr()))
    acc = acc.cpu().numpy()
    return acc[0], acc[1]


class _WholeMatrixPCA:
    """Shared body of the two non-batch PCA estimators (reference estimators.py:84-160): fit on the whole ``[N, d]``
    matrix, then the reference's post-processing (:96-118 / :140-157): projected standard deviations (ddof = 0),
    components sorted by them, ``total_var = X.var(axis=0).sum()``, ``mean_ = X.mean(axis=0)``.

    Components come from the subclass's ``_components(Xd)`` (device rows in, ``[k, d]`` float32 device tensor out);
    the post-processing is two more passes on the device: ``gs_column_moments`` (per-feature sum / sum of squares)
    and ``X @ components^T`` on the f32 MFMA (``gs_linear_forward``) followed by the ``[N, k]`` column statistics.
    """

    GRAM_SIDE_MAX_FEATURES = 8192
    ROWS_PER_LAUNCH = 1 << 20

    def __init__(self, n_components, device=None):
        self.n_components = int(n_components)
        self.batch_support = False
        self._device = device
        self._lib = _lib.load()
        self.transformer = SimpleNamespace()
        self.stdev = np.zeros((self.n_components,))
        self.total_var = 0.0

    # -- building blocks shared by the subclasses ------------------------------------------------------------
    def _device_rows(self, X):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("ganspace_amd needs a HIP device (torch.cuda.is_available() is False)")
        helper = _DeviceIncrementalPCA(self.n_components, _lib.GS_MODE_EXACT, self._device)
        Xd = helper._as_device_rows(X).contiguous()
        self._device = Xd.device
        return Xd

    def _exact_components(self, Xd, centre):
        """Leading eigenvectors of the centred scatter (``centre``: sklearn ``PCA(svd_solver='full')``) or of
        ``X^T X`` (what ``fbpca.pca(raw=True)`` returns when it falls back to a dense SVD).  d <= 8192: one pass of the
        Gram kernel + one top-k eigensolve; wider: the small side of the matrix (rows x rows), one block."""
        torch = _torch()
        n, d = Xd.shape
        k = self.n_components
        if d <= self.GRAM_SIDE_MAX_FEATURES:
            inc = _DeviceIncrementalPCA(k, _lib.GS_MODE_EXACT, self._device)
            for lo in range(0, n, self.ROWS_PER_LAUNCH):
                inc.partial_fit(Xd[lo:lo + self.ROWS_PER_LAUNCH])
            if centre:
                comp = torch.from_numpy(np.array(inc.components_, dtype=np.float32))
            else:
                # uncentred second moment  X^T X = S + n mean mean^T  -> same eigensolver through a second handle
                state = inc.export_state().cpu().numpy()
                mean, S = state[1:1 + d], state[1 + d:].reshape(d, d)
                raw = _DeviceIncrementalPCA(k, _lib.GS_MODE_EXACT, self._device)
                raw._ensure(d)
                st = state.copy()
                st[1 + d:] = (S + n * np.outer(mean, mean)).ravel()
                st[1:1 + d] = 0.0
                raw.import_state(torch.from_numpy(st).to(Xd.device), d)
                comp = torch.from_numpy(np.array(raw.components_, dtype=np.float32))
                raw.close()
            inc.close()
            return comp.to(Xd.device)
        if not centre:
            raise RuntimeError(
                f"fbpca's dense fall-back (l >= rows / 1.25 or l >= feat_dim / 1.25) on feat_dim = {d} > "
                f"{self.GRAM_SIDE_MAX_FEATURES} is not available on the device; use a larger sample")
        # feat_dim >> rows: exact PCA of the block from the small side (T = Xc Xc^T is rows x rows) - the first block of
        # the small-side recurrence IS the plain centred PCA of that block
        inc = _DeviceIncrementalPCA(k, _lib.GS_MODE_SMALLSIDE, self._device)
        try:
            inc.partial_fit(Xd)
        except _lib.GanspaceHipError as e:
            if e.code == _lib.GS_ENOTIMPL:
                raise RuntimeError(
                    f"whole-matrix PCA of a [{n}, {d}] matrix: feat_dim > {self.GRAM_SIDE_MAX_FEATURES} is solved from "
                    "the rows x rows side, which holds n_components + rows + 1 <= 16384; use the batch estimator "
                    "'ipca' (small-side recurrence, any number of rows) for more samples") from e
            raise
        comp = torch.from_numpy(np.array(inc.components_, dtype=np.float32)).to(Xd.device)
        inc.close()
        return comp

    def _components(self, Xd):
        raise NotImplementedError

    def fit(self, X):
        torch = _torch()
        Xd = self._device_rows(X)
        n, d = Xd.shape
        k = self.n_components
        d_true = d
        if d % 4 != 0:
            # the moment / projection kernels read float4 rows: zero-pad the features to a multiple of 4.  A constant-zero
            # feature has zero mean, zero variance and zero weight in every component (centred or not), so the padded
            # problem IS the original one; the reference estimators accept any width (estimators.py:84-160)
            d = d + (4 - d % 4)
            Xp = torch.zeros((n, d), dtype=Xd.dtype, device=Xd.device)
            Xp[:, :d_true] = Xd
            Xd = Xp
        self._d_true = d_true
        comp = self._components(Xd).contiguous()                     # [k, d] float32 on the device
        s1, s2 = _column_moments(self._lib, Xd)
        mean = s1 / n
        self.total_var = float(np.sum(s2 / n - mean * mean))        # X.var(axis=0).sum()
        proj = torch.empty((n, k), dtype=torch.float32, device=Xd.device)
        _lib.check(self._lib.gs_linear_forward(C.c_void_p(Xd.data_ptr()), C.c_void_p(comp.data_ptr()), C.c_void_p(0),
                                               C.c_void_p(proj.data_ptr()), n, d, k, _lib.current_stream_ptr()))
        self.stdev = proj.double().std(dim=0, unbiased=False).cpu().numpy()   # np.dot(components_, X.T).std(axis=1)
        order = np.argsort(self.stdev)[::-1]                         # estimators.py:103-106
        self.stdev = self.stdev[order]
        self.transformer.components_ = comp.cpu().numpy()[order][:, :d_true].astype(np.float32)
        self.transformer.mean_ = mean[None, :d_true].astype(np.float32)
        c64 = self.transformer.components_.astype(np.float64)
        gram = c64 @ c64.T
        off = np.abs(gram - np.diag(np.diag(gram))).max() if k > 1 else 0.0
        if off > 1e-4:
            print("PCA components not orthogonal, max dot", off)

    def get_components(self):
        var_ratio = self.stdev ** 2 / self.total_var
        return self.transformer.components_, self.stdev, var_ratio


class PCAEstimator(_WholeMatrixPCA):
    """Drop-in for the reference ``PCAEstimator`` (estimators.py:84-118): sklearn ``PCA(k, svd_solver='full')``."""

    def __init__(self, n_components, device=None):
        super().__init__(n_components, device=device)
        self.solver = "full"

    def get_param_str(self):
        return f"pca-{self.solver}_c{self.n_components}"

    def _components(self, Xd):
        return self._exact_components(Xd, centre=True)


class FacebookPCAEstimator(_WholeMatrixPCA):
    """Drop-in for the reference ``FacebookPCAEstimator`` (estimators.py:124-160):
    ``fbpca.pca(X, k, n_iter=2, raw=True, l=2k)`` - the randomized range finder with two normalised power iterations -
    as ``gs_randomized_pca`` (csrc/gs_rangefinder.hip).  The test matrix is drawn here from NumPy's global stream with
    the very call fbpca makes (``np.random.uniform(-1, 1, (n, l))`` for rows >= feat_dim, ``(l, m)`` otherwise), so a
    given RNG state gives the matrix the reference would use.  Where fbpca falls back to a dense SVD
    (``l >= rows / 1.25 or l >= feat_dim / 1.25``) the exact solver runs."""

    def __init__(self, n_components, device=None):
        super().__init__(n_components, device=device)
        self.n_iter = 2
        self.l = 2 * self.n_components

    def get_param_str(self):
        return "fbpca_c{}_it{}_l{}".format(self.n_components, self.n_iter, self.l)

    def _components(self, Xd):
        torch = _torch()
        n, d = Xd.shape
        k, l = self.n_components, self.l
        if l >= n / 1.25 or l >= getattr(self, "_d_true", d) / 1.25:
            return self._exact_components(Xd, centre=False)
        if l > 256:
            raise RuntimeError(f"fbpca on the device handles l = 2 * n_components <= 256 (got {l})")
        # (features zero-padded to a multiple of 4 by fit(): the draw has the shape fbpca would use for the TRUE width, the
        #  padded rows of the test matrix are zero - they only ever multiply zero columns)
        d_true = getattr(self, "_d_true", d)
        omega = np.random.uniform(low=-1.0, high=1.0, size=(d_true, l) if n >= d_true else (l, n))
        if n >= d_true and d_true != d:
            omega = np.concatenate([omega, np.zeros((d - d_true, l))], axis=0)
        om = torch.from_numpy(omega).to(Xd.device)
        comp = torch.empty((k, d), dtype=torch.float32, device=Xd.device)
        sv = torch.empty(k, dtype=torch.float64, device=Xd.device)
        _lib.check(self._lib.gs_randomized_pca(C.c_void_p(Xd.data_ptr()), n, d, k, l, self.n_iter,
                                               C.c_void_p(om.data_ptr()), C.c_void_p(comp.data_ptr()),
                                               C.c_void_p(sv.data_ptr()), _lib.current_stream_ptr()))
        self.singular_values_ = sv.cpu().numpy()
        return comp


def get_estimator(name, n_components, alpha=1.0):
    """Factory with the reference's names and error behaviour (estimators.py:206-218)."""
    if name == "pca":
        return PCAEstimator(n_components)
    if name == "ipca":
        return IPCAEstimator(n_components, "faithful")
    if name == "ipca-exact":
        return IPCAEstimator(n_components, "exact")
    if name == "fbpca":
        return FacebookPCAEstimator(n_components)
    if name in ("ica", "spca"):
        # not PCA, not batchable, not on the accelerated path: the reference's own scikit-learn arithmetic
        from . import cpu_estimators
        return cpu_estimators.make(name, n_components, alpha)
    raise RuntimeError("Unknown estimator")

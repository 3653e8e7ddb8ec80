"""The reference-side binding a maintainer of harskish/ganspace would add to ``estimators.py``:
a ctypes stub over ``include/ganspace_hip.h`` that satisfies the duck-typed estimator protocol
``decomposition.compute()`` drives (``estimators.py:55-81``: ``batch_support``, ``fit_partial``,
``get_components``, ``get_param_str``, ``transformer.mean_``).  It deliberately does NOT import ``ganspace_amd``:
it binds the C ABI directly, which is what the boundary is for.  Executed by
``tests/test_gpu_decomposition.py::test_integration_stub_runs_against_the_c_abi``.
"""
# estimators.py (reference side) -- ctypes stub over include/ganspace_hip.h
import ctypes as C, numpy as np, torch

import os
_gs = C.CDLL(os.environ.get("GANSPACE_HIP_LIB", "libganspace_hip.so"))   # torch must already be imported (shared HIP runtime)
_gs.gs_ipca_create.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
_gs.gs_ipca_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
_gs.gs_ipca_finalize.argtypes = [C.c_void_p] * 9
_gs.gs_last_error.restype = C.c_char_p

class HipIPCAEstimator():
    def __init__(self, n_components):
        self.n_components, self.whiten, self.batch_support = n_components, False, True
        self._h, self._d, self._res = C.c_void_p(), None, None
        self.transformer = self                      # decomposition.py:289 reads .transformer.mean_

    def get_param_str(self):
        return "ipca_c{}".format(self.n_components)

    def fit_partial(self, X):                        # X: float32 [NB, d] host array (decomposition.py:264)
        if self._d is None:
            self._d = X.shape[1]
            rc = _gs.gs_ipca_create(self._d, self.n_components, 1, 0, torch.cuda.current_device(), C.byref(self._h))
            assert rc == 0, _gs.gs_last_error()
        Xd = torch.from_numpy(X).cuda()              # the reference hands over a host buffer
        rc = _gs.gs_ipca_update(self._h, Xd.data_ptr(), X.shape[0], X.shape[1],
                                torch.cuda.current_stream().cuda_stream)
        if rc == -1:                                 # GS_EINVAL == sklearn's ValueError
            print('\nIPCA error:', _gs.gs_last_error().decode()); return False
        assert rc == 0, _gs.gs_last_error()
        self._res = None
        return True

    def _finalize(self):
        if self._res is None:
            k, d = self.n_components, self._d
            comp = np.empty((k, d), np.float32); sv, ev, evr = (np.empty(k) for _ in range(3))
            mean, var = np.empty(d), np.empty(d); n = C.c_int64()
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            rc = _gs.gs_ipca_finalize(self._h, p(comp), p(sv), p(mean), p(var), p(ev), p(evr),
                                      C.cast(C.byref(n), C.c_void_p), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, _gs.gs_last_error()
            self._res = dict(components_=comp, mean_=mean, explained_variance_=ev, explained_variance_ratio_=evr)
        return self._res

    mean_ = property(lambda self: self._finalize()["mean_"])

    def get_components(self):
        r = self._finalize()
        return r["components_"], np.sqrt(r["explained_variance_"]), r["explained_variance_ratio_"]

"""Multi-rank merge of the sklearn-faithful estimators on the device (``gs_ipca_lowrank_export`` /
``gs_ipca_lowrank_merge``) and the torch-free collective entry ``gs_ipca_allreduce`` (RCCL through the C ABI).

Checker: ``oracle.ipca.merge_lowrank_states`` - one more step of sklearn's recurrence
(``_incremental_pca.py:335-378``) on the stacked per-rank states (second bullet)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ipca as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    return torch.device("cuda", 0)


def _blocks(n_blocks, rows, d, latent, seed, decay=1.15):
    rs = np.random.RandomState(seed)
    A = rs.standard_normal((latent, d)) * (decay ** -np.arange(latent))[:, None] * 3.0
    mu = rs.standard_normal(d) * 0.5
    return [(rs.standard_normal((rows, latent)) @ A + mu + 0.05 * rs.standard_normal((rows, d))).astype(np.float32)
            for _ in range(n_blocks)]


def _set_state(orc, m):
    orc.components_, orc.singular_values_ = m["components_"], m["singular_values_"]
    orc.mean_, orc.var_, orc.n_samples_seen_ = m["mean_"], m["var_"], m["n_samples_seen_"]


@pytest.mark.parametrize("d,k,rows,mode", [(96, 12, 400, "faithful"), (512, 20, 1000, "faithful"),
                                           (9000, 16, 300, "smallside")])
def test_lowrank_merge_on_device_matches_oracle_merge(dev, d, k, rows, mode):
    """Three shards of a 12-block stream (one of them long enough to reach the deferred-diagonalisation state):
    exported states equal the per-shard oracle fits, the device merge equals the oracle merge of the same states
    (float64 both: 1e-9), the merged leading components equal the sequential fit, and the merged handle keeps
    fitting (one more block) like the oracle does from the merged state."""
    from ganspace_amd import _lib
    from ganspace_amd.estimators import IPCAEstimator
    blocks = _blocks(13, rows, d, 48, seed=d)
    shards = [blocks[0:7], blocks[7:10], blocks[10:12]]
    states, orc_states = [], []
    for sh in shards:
        e, o = IPCAEstimator(k, mode), O.SklearnRecurrenceOracle(k)
        for X in sh:
            assert e.fit_partial(torch.from_numpy(X).to(dev))
            o.partial_fit(X)
        st = e.transformer.export_lowrank()
        assert st.numel() == e.transformer.lowrank_len()
        states.append(st)
        orc_states.append(O.pack_lowrank_state(o))
        s = st.cpu().numpy()
        assert s[0] == o.n_samples_seen_
        np.testing.assert_allclose(s[1:1 + d], o.mean_, atol=2e-6)
        np.testing.assert_allclose(s[1 + d:1 + 2 * d], o.var_ * o.n_samples_seen_, rtol=1e-4)
        np.testing.assert_allclose(np.sqrt(s[1 + 2 * d:1 + 2 * d + k]), o.singular_values_, rtol=2e-4)
        assert O.signed_cosines(s[1 + 2 * d + k:].reshape(k, d), o.components_).min() > 1 - 5e-6
    # a rank that saw nothing joins with zeros
    empty = IPCAEstimator(k, mode)
    empty.transformer._ensure(d)
    states.append(empty.transformer.export_lowrank())
    assert float(states[-1].abs().sum()) == 0.0
    stacked = torch.stack(states)
    merged = IPCAEstimator(k, mode)
    merged.transformer._ensure(d)
    merged.transformer.merge_lowrank(stacked)
    ref = O.merge_lowrank_states([s.cpu().numpy() for s in states], k, d)
    t = merged.transformer
    assert int(t.n_samples_seen_) == ref["n_samples_seen_"] == 12 * rows
    cos = O.signed_cosines(t.components_, ref["components_"])
    assert cos.min() > 1 - 1e-6, cos                      # float32 output of a float64 merge
    np.testing.assert_allclose(t.singular_values_, ref["singular_values_"], rtol=1e-9)
    np.testing.assert_allclose(t.mean_, ref["mean_"], atol=1e-12)
    np.testing.assert_allclose(t.var_, ref["var_"], rtol=1e-12)
    np.testing.assert_allclose(t.explained_variance_ratio_, ref["explained_variance_ratio_"], rtol=1e-9)
    # against the sequential single-process fit: leading components (the truncation order differs)
    seq = O.SklearnRecurrenceOracle(k)
    for X in blocks[:12]:
        seq.partial_fit(X)
    assert O.signed_cosines(t.components_, seq.components_)[:k // 2].min() > 0.9999
    np.testing.assert_allclose(t.mean_, seq.mean_, atol=2e-6)
    np.testing.assert_allclose(t.var_, seq.var_, rtol=1e-4)
    # the merged handle is a normal estimator again
    cont = O.SklearnRecurrenceOracle(k)
    _set_state(cont, O.merge_lowrank_states(orc_states, k, d))
    cont.partial_fit(blocks[12])
    assert merged.fit_partial(torch.from_numpy(blocks[12]).to(dev))
    assert O.signed_cosines(merged.get_components()[0], cont.components_)[:k // 2].min() > 1 - 1e-5
    np.testing.assert_allclose(merged.transformer.singular_values_[:k // 2], cont.singular_values_[:k // 2], rtol=1e-3)


# ---- gs_ipca_allreduce: RCCL through the C ABI, no torch.distributed -------------------------------------------
class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _rccl():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = C.CDLL(path if os.path.exists(path) else "librccl.so.1", mode=C.RTLD_GLOBAL)
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib


@pytest.mark.parametrize("mode", ["exact", "faithful", "smallside"])
def test_gs_ipca_allreduce_single_rank_rccl_communicator(dev, mode):
    """A one-rank ncclComm_t created by the *caller* (ctypes on RCCL, as a non-torch host would) handed to
    ``gs_ipca_allreduce``: export -> collectives -> re-centre / merge -> import must leave the fit unchanged."""
    from ganspace_amd import _lib
    from ganspace_amd.estimators import IPCAEstimator
    lib, nccl = _lib.load(), _rccl()
    uid, comm = _UniqueId(), C.c_void_p()
    assert nccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert nccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        d, k = (9000, 12) if mode == "smallside" else (256, 12)
        est = IPCAEstimator(k, mode)
        for X in _blocks(7, 500, d, 40, seed=3):
            assert est.fit_partial(torch.from_numpy(X).to(dev))
        before = est.transformer.components_.copy()
        sv, mean, var = (est.transformer.singular_values_.copy(), est.transformer.mean_.copy(), est.transformer.var_.copy())
        _lib.check(lib.gs_ipca_allreduce(est.transformer._h, comm, _lib.current_stream_ptr()))
        est.transformer._cache = None
        cos = O.signed_cosines(est.transformer.components_, before)
        assert cos.min() > 1 - 1e-7, cos
        np.testing.assert_allclose(est.transformer.singular_values_, sv, rtol=2e-6)
        np.testing.assert_allclose(est.transformer.mean_, mean, atol=1e-9)
        np.testing.assert_allclose(est.transformer.var_, var, rtol=1e-9)
        assert int(est.transformer.n_samples_seen_) == 3500
    finally:
        nccl.ncclCommDestroy(comm)

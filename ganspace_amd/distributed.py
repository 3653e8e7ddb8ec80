"""Multi-GPU sharding of the sample loop (new design; the reference is single-process,
SURVEY.md §2.2/§8e).

The N-sample loop of ``decomposition.compute`` (decomposition.py:245-267) is embarrassingly
parallel in the rows: PCA sufficient statistics are additive.  One process per GPU
(``torch.distributed``, backend ``nccl`` = RCCL over xGMI); rank r owns a contiguous range
of the IPCA blocks, accumulates a local (n, mean, centred scatter) and the ranks meet in
ONE exchange step before the eigensolve:

    all-reduce [ n | n*mean ]            (d+1 float64: latency-bound)
    re-centre the local scatter about the global mean (Chan et al. pairwise merge)
    all-reduce the packed upper triangle of C   (d(d+1)/2 float64: 1 MiB at d=512)

The sklearn-faithful estimators (``ipca``: Gram side and small side) cannot add their states - the recurrence truncates
to k components after every block - so each rank runs it on its own blocks and the ranks meet in one all-gather of the
low-rank states ``(n, mean, m2, S^2, V)`` followed by a merge solve (second bullet).

No collective sits on the data path itself.  A rank that owns no block (more ranks than blocks) contributes an
all-zero state instead of stalling the others.  No N > 1 RCCL run has been measured in the build environment
(``gpurun`` exposes one GPU); the logic is covered by world_size-2 gloo tests and a 1-rank RCCL test.
"""
from __future__ import annotations

import ctypes as C

from . import _lib


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced ``[lo, hi)`` share of ``n_items`` for ``rank`` (first ranks get the
    remainder), so a ``world``-GPU run consumes exactly the blocks of the 1-GPU run."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _recenter(state, d, new_mean, n_local=None):
    """``C += n (mean-new)(mean-new)^T ; mean = new`` in place on a state vector.  ``n_local`` = the state's sample
    count when the caller knows it on the host (saves a device round trip)."""
    import torch
    if (float(state[0]) if n_local is None else n_local) <= 0:
        state[1:1 + d] = new_mean.to(state.dtype)       # an empty state has no scatter to move
        return state
    if state.is_cuda:
        lib = _lib.load()
        new_mean = new_mean.to(device=state.device, dtype=torch.float64).contiguous()
        _lib.check(lib.gs_state_recenter(C.c_void_p(state.data_ptr()), d, C.c_void_p(new_mean.data_ptr()),
                                         _lib.current_stream_ptr()))
    else:
        # host tensors only occur in the gloo unit tests of the collective logic
        delta = state[1:1 + d] - new_mean
        state[1 + d:].view(d, d).add_(torch.outer(delta, delta) * state[0])
        state[1:1 + d] = new_mean
    return state


def merge_states(states, d):
    """Merge exported states ``[n | mean | C]`` in one process (resume / tests)."""
    import torch
    n = sum(s[0] for s in states)
    mean = sum(s[0] * s[1:1 + d] for s in states) / n
    out = torch.zeros_like(states[0])
    for s in states:
        s = _recenter(s.clone(), d, mean)
        out[1 + d:] += s[1 + d:]
    out[0] = n
    out[1:1 + d] = mean
    return out


PACK_MAX_FEATURES = 1024     # beyond this the index tensors of the packing (two int64 per packed element, cached per
                             # (d, device): 8 MB at d = 1024, 134 MB at 4096) cost more than the bytes they save


_TRIU = {}


def _triu_flat(d, device):
    """Flat indices (into the row-major d*d buffer) of the upper triangle and of its mirror image, cached."""
    import torch
    key = (int(d), str(device))
    if key not in _TRIU:
        iu = torch.triu_indices(d, d, device=device)
        _TRIU[key] = (iu[0] * d + iu[1], iu[1] * d + iu[0])
    return _TRIU[key]


def pack_upper(scatter, d):
    """Row-major upper triangle (diagonal included) of a symmetric ``d x d`` matrix stored flat: d(d+1)/2 values."""
    up, _ = _triu_flat(d, scatter.device)
    return scatter.reshape(-1)[up]


def unpack_upper(packed, d, out):
    """Inverse of :func:`pack_upper` into the flat ``d*d`` buffer ``out`` (both triangles filled)."""
    up, lo = _triu_flat(d, packed.device)
    flat = out.reshape(-1)
    flat[lo] = packed
    flat[up] = packed
    return out


def allreduce_state(state, d, group=None, n_local=None):
    """In-place global merge of every rank's exported state: a sum all-reduce of ``[n | n mean]``, the Chan
    re-centring about the global mean, then a sum all-reduce of the (packed) centred scatter.  ``n_local``: this
    rank's sample count if known on the host (no device round trip then)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return state
    hdr = torch.cat([state[:1], state[0] * state[1:1 + d]])
    dist.all_reduce(hdr, op=dist.ReduceOp.SUM, group=group)
    if n_local is None or n_local <= 0:
        if float(hdr[0]) <= 0:
            return state                          # nobody has seen a sample
    mean = hdr[1:] / hdr[0]
    _recenter(state, d, mean, n_local)
    scatter = state[1 + d:]
    if d <= PACK_MAX_FEATURES:
        packed = pack_upper(scatter, d)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        unpack_upper(packed, d, scatter)
    else:
        dist.all_reduce(scatter, op=dist.ReduceOp.SUM, group=group)
    state[0] = hdr[0]
    return state


def gather_lowrank_states(state, group=None):
    """All-gather every rank's low-rank state ``[n | mean | m2 | lam | V]`` (float64, same length on every rank):
    returns ``[world, len]``.  The payload is k x d per rank (42-86 MB at d = 131 072, k = 80): one ring all-gather
    over the xGMI links, once per job."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return state[None, :]
    world = dist.get_world_size(group)
    parts = [torch.empty_like(state) for _ in range(world)]
    dist.all_gather(parts, state.contiguous(), group=group)
    return torch.stack(parts)


def allreduce_estimator(estimator, group=None, d=None):
    """Merge the state of ``estimator`` across all ranks, in place.  ``d`` (the feature count every rank agrees on)
    lets a rank that fitted nothing create its handle and contribute an empty state.

    ``ipca-exact``  additive statistics: two all-reduces (n / mean, then the centred scatter) before the eigensolve.
    ``ipca``        the sklearn-faithful recurrence is sequential in the blocks, so every rank ran it on ITS blocks;
                    the low-rank states are all-gathered and merged by one more step of the same recurrence with each
                    state as a pre-compressed batch (``gs_ipca_lowrank_merge``) - not identical to a single sequential
                    fit (different truncation order; compare the leading components), identical on every rank."""
    t = estimator.transformer
    if t._h is None:
        if d is None:
            raise RuntimeError("allreduce_estimator: this rank fitted no block; pass d= so that it can join with "
                               "an empty state")
        t._ensure(int(d))
    if t._mode == _lib.GS_MODE_EXACT:
        st = allreduce_state(t.export_state(), t._d, group, n_local=getattr(t, "_n_host", None))
        t.import_state(st, t._d)
    else:
        t.merge_lowrank(gather_lowrank_states(t.export_lowrank(), group))
    return estimator

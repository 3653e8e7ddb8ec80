"""Multi-GPU sharding of the sample loop (new design; the reference is single-process,
SURVEY.md §2.2/§8e).

The N-sample loop of ``decomposition.compute`` (decomposition.py:245-267) is embarrassingly
parallel in the rows: PCA sufficient statistics are additive.  One process per GPU
(``torch.distributed``, backend ``nccl`` = RCCL over xGMI); rank r owns a contiguous range
of the IPCA blocks, accumulates a local (n, mean, centred scatter) and the ranks meet in
ONE exchange step before the eigensolve:

    all-reduce [ n | n*mean ]            (d+1 float64: latency-bound)
    re-centre the local scatter about the global mean (Chan et al. pairwise merge)
    all-reduce C                         (d*d float64: 2 MiB at d=512)

No collective sits on the data path itself.
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


def _recenter(state, d, new_mean):
    """``C += n (mean-new)(mean-new)^T ; mean = new`` in place on a state vector."""
    import torch
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


def allreduce_state(state, d, group=None):
    """In-place global merge of every rank's exported state (two sum all-reduces)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return state
    hdr = torch.cat([state[:1], state[0] * state[1:1 + d]])
    dist.all_reduce(hdr, op=dist.ReduceOp.SUM, group=group)
    mean = hdr[1:] / hdr[0]
    _recenter(state, d, mean)
    scatter = state[1 + d:]
    dist.all_reduce(scatter, op=dist.ReduceOp.SUM, group=group)
    state[0] = hdr[0]
    return state


def allreduce_estimator(estimator, group=None):
    """Merge the EXACT-mode state of ``estimator`` across all ranks, in place."""
    t = estimator.transformer
    st = allreduce_state(t.export_state(), t._d, group)
    t.import_state(st, t._d)
    return estimator

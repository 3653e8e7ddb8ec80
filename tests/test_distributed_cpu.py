"""world_size-2 gloo tests (CPU) of the multi-GPU merge: shard plan + the two-all-reduce Chan
merge of (n, mean, centred scatter) equals the statistics of the concatenated data."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ganspace_amd import distributed as D


def _state_of(X):
    X = np.asarray(X, dtype=np.float64)
    n, d = X.shape
    mean = X.mean(0)
    Xc = X - mean
    return torch.from_numpy(np.concatenate([[n], mean, (Xc.T @ Xc).ravel()]))


def _data(rank_rows, d=24, seed=0):
    rs = np.random.RandomState(seed)
    return [rs.standard_normal((r, d)) * rs.uniform(0.5, 2, d) + rs.uniform(-5, 5, d) for r in rank_rows]


def test_shard_range_is_contiguous_and_balanced():
    for n, w in [(100, 8), (800, 8), (7, 3), (5, 8), (1, 1)]:
        spans = [D.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_merge_states_matches_concatenation():
    parts = _data([300, 50, 1000])
    merged = D.merge_states([_state_of(p) for p in parts], 24)
    full = _state_of(np.concatenate(parts))
    torch.testing.assert_close(merged, full, rtol=1e-10, atol=1e-9)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    parts = _data([400, 900], seed=3)
    st = _state_of(parts[rank])
    D.allreduce_state(st, 24)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), st.numpy())
    dist.destroy_process_group()


def test_allreduce_state_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    full = _state_of(np.concatenate(_data([400, 900], seed=3))).numpy()
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npy")
        np.testing.assert_allclose(got, full, rtol=1e-10, atol=1e-9)


def _worker_single(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = _state_of(_data([321], seed=9)[0])
    ref = st.clone()
    D.allreduce_state(st, 24)
    np.save(os.path.join(out_dir, "single.npy"), (st - ref).abs().max().numpy())
    dist.destroy_process_group()


def test_allreduce_with_one_rank_is_a_fixed_point(tmp_path):
    mp.spawn(_worker_single, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert float(np.load(tmp_path / "single.npy")) < 1e-9


def test_allreduce_is_identity_without_process_group():
    st = _state_of(_data([50])[0])
    assert D.allreduce_state(st.clone(), 24).equal(st)

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


def test_pack_unpack_upper_roundtrip():
    rs = np.random.RandomState(1)
    for d in (1, 5, 24, 129):
        A = rs.standard_normal((d, d))
        S = torch.from_numpy(A + A.T).reshape(-1).clone()
        packed = D.pack_upper(S, d)
        assert packed.numel() == d * (d + 1) // 2
        out = torch.empty_like(S)
        D.unpack_upper(packed, d, out)
        assert out.equal(S)


def _worker_empty_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # more ranks than blocks: rank 1 saw nothing and contributes an all-zero state
    st = _state_of(_data([700], seed=5)[0]) if rank == 0 else torch.zeros(1 + 24 + 24 * 24, dtype=torch.float64)
    D.allreduce_state(st, 24)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), st.numpy())
    dist.destroy_process_group()


def test_allreduce_with_an_empty_rank(tmp_path):
    mp.spawn(_worker_empty_rank, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    full = _state_of(_data([700], seed=5)[0]).numpy()
    for r in range(2):
        np.testing.assert_allclose(np.load(tmp_path / f"rank{r}.npy"), full, rtol=1e-10, atol=1e-9)


# ---- the rank-sharded compute() plan: same blocks, same z rows as the single-process run -----------------------------

@pytest.mark.parametrize("n,batch,k,world", [(1_000_000, 10_000, 80, 8), (8_000_000, 10_000, 80, 8), (10_000, 512, 20, 2),
                                             (12_000, 1000, 10, 3), (4000, 500, 10, 5), (2100, 100, 8, 4)])
def test_shard_plan_union_is_the_single_rank_block_list(n, batch, k, world):
    from ganspace_amd.decomposition import _Plan
    plan = _Plan.make(n, batch, k)
    single = list(plan.block_starts)
    shards = [plan.shard_blocks(r, world) for r in range(world)]
    assert sum(shards, []) == single                          # contiguous shares, in order, nothing lost
    assert max(map(len, shards)) - min(map(len, shards)) <= 1
    per_block = -(-plan.NB // plan.B) * plan.B
    for starts in shards:
        lo, hi = plan.batch_span(starts)
        if not starts:
            assert (lo, hi) == (0, 0)
            continue
        # every row a block of this share reads lies inside the generated batches, and inside n_lat
        assert lo * plan.B <= starts[0] and starts[-1] + per_block <= hi * plan.B <= plan.n_lat


def test_sharded_presample_reads_the_same_z_rows():
    """Every rank draws the whole seed list (models/wrappers.py:168-169) and generates only its batches: the rows it
    holds are bit-identical to the single-process array and the global stream ends in the same state."""
    from ganspace_amd.decomposition import SEED_SAMPLING, _Plan, _presample
    from ganspace_amd.wrappers import StyleGAN2
    cpu = torch.device("cpu")
    model = StyleGAN2(cpu, "cat")                 # Z space: latent_from_z is the identity, nothing touches the GPU
    shape = (1, 512)
    plan = _Plan.make(6000, 250, 10)              # NB = 2000, 3 blocks, 8 batches per block
    np.random.seed(SEED_SAMPLING)
    full, row0 = _presample(model, plan, shape, cpu)
    assert row0 == 0 and full.shape[0] == plan.n_lat
    state_after = np.random.get_state()[1].copy()
    world = 2
    for rank in range(world):
        starts = plan.shard_blocks(rank, world)
        lo, hi = plan.batch_span(starts)
        np.random.seed(SEED_SAMPLING)
        part, r0 = _presample(model, plan, shape, cpu, lo, hi)
        assert r0 == lo * plan.B and part.shape[0] == (min(hi, plan.n_lat // plan.B) - lo) * plan.B
        assert torch.equal(part, full[r0:r0 + part.shape[0]])
        for gi in starts:                         # the rows block gi consumes (decomposition.py:246-247)
            assert torch.equal(part[gi - r0:gi - r0 + plan.NB], full[gi:gi + plan.NB])
        np.testing.assert_array_equal(np.random.get_state()[1], state_after)


# ---- the sklearn-faithful estimator sharded: per-rank recurrence + low-rank merge (second bullet) ----
def _faithful_blocks(n_blocks=12, rows=400, d=96, latent=40, seed=5):
    rs = np.random.RandomState(seed)
    A = rs.standard_normal((latent, d)) * (1.18 ** -np.arange(latent))[:, None] * 3.0
    mu = rs.standard_normal(d)
    return [(rs.standard_normal((rows, latent)) @ A + mu + 0.05 * rs.standard_normal((rows, d))).astype(np.float32)
            for _ in range(n_blocks)]


def _faithful_worker(rank, world, port, out_dir):
    from oracle import ipca as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, blocks = 12, _faithful_blocks()
    lo, hi = D.shard_range(len(blocks), rank, world)
    orc = O.SklearnRecurrenceOracle(k)
    for X in blocks[lo:hi]:
        orc.partial_fit(X)
    st = torch.from_numpy(O.pack_lowrank_state(orc))
    allst = D.gather_lowrank_states(st)                     # the product's collective (gloo here, RCCL on the GPUs)
    assert allst.shape == (world, st.numel())
    np.save(os.path.join(out_dir, f"states{rank}.npy"), allst.numpy())
    dist.destroy_process_group()


def test_faithful_lowrank_merge_world_size_2_gloo(tmp_path):
    """Two ranks run the recurrence on their halves of the block list, all-gather the low-rank states through
    ``distributed.gather_lowrank_states`` and merge (oracle restatement of ``gs_ipca_lowrank_merge``): every rank holds
    the same states, and the merged leading components equal those of the sequential single-process fit."""
    from oracle import ipca as O
    port = _free_port()
    mp.spawn(_faithful_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0, s1 = np.load(tmp_path / "states0.npy"), np.load(tmp_path / "states1.npy")
    np.testing.assert_array_equal(s0, s1)
    k, d, blocks = 12, 96, _faithful_blocks()
    merged = O.merge_lowrank_states(list(s0), k, d)
    seq = O.SklearnRecurrenceOracle(k)
    for X in blocks:
        seq.partial_fit(X)
    assert merged["n_samples_seen_"] == seq.n_samples_seen_ == 12 * 400
    np.testing.assert_allclose(merged["mean_"], seq.mean_, atol=1e-12)
    np.testing.assert_allclose(merged["var_"], seq.var_, rtol=1e-10)           # Chan merge is exact
    cos = O.signed_cosines(merged["components_"], seq.components_)
    assert cos[:8].min() > 0.9999, cos        # truncation order differs: leading components only
    np.testing.assert_allclose(merged["singular_values_"][:8], seq.singular_values_[:8], rtol=1e-3)
    # and against the exact PCA of all rows
    ex = O.exact_pca(blocks, k)
    assert O.signed_cosines(merged["components_"], ex["components_"])[:8].min() > 0.9999


def test_lowrank_merge_of_one_state_is_the_identity():
    from oracle import ipca as O
    k, d = 6, 40
    orc = O.SklearnRecurrenceOracle(k)
    for X in _faithful_blocks(4, 200, d, 20, seed=9):
        orc.partial_fit(X)
    st = O.pack_lowrank_state(orc)
    empty = np.zeros_like(st)
    m = O.merge_lowrank_states([st, empty], k, d)
    assert O.signed_cosines(m["components_"], orc.components_).min() > 1 - 1e-12
    np.testing.assert_allclose(m["singular_values_"], orc.singular_values_, rtol=1e-12)
    np.testing.assert_allclose(m["var_"], orc.var_, rtol=1e-12)

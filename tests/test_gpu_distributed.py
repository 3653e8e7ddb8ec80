"""The SHARDED product path, executed: ``get_or_compute`` and ``bench.py`` with world_size 2 on the one GPU a test box
has (gloo backend, device tensors), against the single-process run of the same configuration.

What an 8-GPU RCCL run executes - ``_Plan.shard_blocks``, per-rank z generation (``_presample(batch_lo, batch_hi)``),
``allreduce_estimator`` (exact: n / mean / packed scatter; ``ipca``: all-gather of the low-rank states + merge solve),
the head broadcast for ``random_stdevs``, the sharded regression with all-reduced normal equations, the rank-0 write -
runs here with two ranks; only the transport differs (gloo instead of RCCL over xGMI)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import ipca as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(argv, world, extra_env=None, timeout=900):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for rc, out, err in outs:
        assert rc == 0, f"rank failed (rc={rc})\n--- stdout ---\n{out[-3000:]}\n--- stderr ---\n{err[-6000:]}"
    return outs


def _run_case(case, world, run_dir):
    os.makedirs(run_dir, exist_ok=True)
    outs = _launch([WORKER, case, str(run_dir)], world)
    line = [l for l in outs[0][1].splitlines() if l.startswith("WROTE ")]
    assert line, outs[0][1][-2000:]
    return np.load(line[-1].split(" ", 1)[1], allow_pickle=False)


def _compare(one, two, k_tight, cos_tol, lat=True, rtol=1e-4):
    k = one["act_comp"].shape[0]
    c = O.signed_cosines(two["act_comp"].reshape(k, -1), one["act_comp"].reshape(k, -1))
    assert c[:k_tight].min() > 1 - cos_tol, c
    np.testing.assert_allclose(two["act_mean"], one["act_mean"], atol=2e-6)
    np.testing.assert_allclose(two["act_stdev"][:k_tight], one["act_stdev"][:k_tight], rtol=rtol)
    np.testing.assert_allclose(two["var_ratio"][:k_tight], one["var_ratio"][:k_tight], rtol=rtol)
    np.testing.assert_allclose(two["random_stdevs"], one["random_stdevs"], rtol=1e-4)   # same head rows (broadcast)
    np.testing.assert_allclose(two["lat_stdev"][:k_tight], one["lat_stdev"][:k_tight], rtol=1e-3)
    if lat:
        lc = O.signed_cosines(two["lat_comp"].reshape(k, -1), one["lat_comp"].reshape(k, -1))
        assert lc[:k_tight].min() > 1 - max(cos_tol, 1e-5), lc


@pytest.mark.parametrize("case,k", [("w_exact", 20), ("z_exact", 10)])
def test_sharded_exact_compute_equals_single_process(tmp_path, case, k):
    """``--est=ipca-exact`` (additive statistics): the two-rank file equals the one-rank file to 1e-6 in signed
    cosine - W-space (cfg2 in miniature) and Z-space (cfg4 in miniature: regression all-reduce, head broadcast)."""
    one = _run_case(case, 1, tmp_path / "one")
    two = _run_case(case, 2, tmp_path / "two")
    assert sorted(one.keys()) == sorted(two.keys())
    _compare(one, two, k, 1e-6)


def test_sharded_faithful_wide_layer_equals_in_process_merge_of_the_same_shards(tmp_path):
    """``--est=ipca`` on a wide layer (BigGAN gen_z, d = 32 768: small-side recurrence per rank) with two ranks against
    the same computation done in THIS process through the library API: presample the one z stream, run rank 0's and
    rank 1's block ranges through two estimators, export, merge.  BigGAN's gen_z activation is affine in z with a flat
    leading spectrum, so a sequential fit and a merged fit legitimately differ (truncation order); the two-rank file
    must equal the in-process merge - plumbing (shard plan, per-rank z batches, all-gather, rank-0 write) is what this
    pins, the merge arithmetic itself is pinned against the oracle in tests/test_gpu_merge.py."""
    import torch
    from ganspace_amd import decomposition as dec
    from ganspace_amd.config import Config
    from ganspace_amd.estimators import get_estimator
    from ganspace_amd.wrappers import get_instrumented_model
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    two = _run_case("wide_ipca", 2, tmp_path / "two")
    kw = dict(dist_worker.CASES["wide_ipca"], output_class=250)
    cfg = Config(**kw)
    dev = torch.device("cuda", 0)
    inst = get_instrumented_model(cfg.model, cfg.output_class, cfg.layer, dev)
    model, layer, k = inst.model, cfg.layer, cfg.components
    inst.retain_layer(layer)
    input_shape = model.get_latent_shape()
    plan = dec._Plan.make(cfg.n, cfg.batch_size, k)
    torch.manual_seed(dec.SEED_SAMPLING)
    np.random.seed(dec.SEED_SAMPLING)
    latents, _ = dec._presample(model, plan, input_shape, dev)
    states, d = [], 32768
    for r in range(2):
        est = get_estimator("ipca", k, 1.0)
        dec._fit_blocks(est, inst, latents, plan, layer, d, False, dev, plan.shard_blocks(r, 2), 0)
        states.append(est.transformer.export_lowrank())
    merged = get_estimator("ipca", k, 1.0)
    merged.transformer._ensure(d)
    merged.transformer.merge_lowrank(torch.stack(states))
    comp, stdev, ratio = merged.get_components()
    c = O.signed_cosines(two["act_comp"].reshape(k, -1), comp)
    assert c.min() > 1 - 1e-6, c
    np.testing.assert_allclose(two["act_stdev"], stdev, rtol=1e-6)
    np.testing.assert_allclose(two["var_ratio"], ratio, rtol=1e-6)
    np.testing.assert_allclose(two["act_mean"].ravel(), merged.transformer.mean_, atol=1e-6)
    assert two["lat_comp"].shape == (k, 1, 128)            # the sharded regression ran (Z-space layer)
    inst.close()


@pytest.mark.parametrize("case,k", [("w_ipca", 20)])
def test_sharded_faithful_compute_matches_single_process_on_leading_components(tmp_path, case, k):
    """``--est=ipca`` (the reference's default, sequential in the blocks): each rank runs the sklearn-faithful
    recurrence on its share, the ranks all-gather ``(n, mean, m2, S, V)`` and merge by one more step of the same
    recurrence on the stacked low-rank states.  The truncation order differs from the sequential fit, so only the
    leading components are compared: signed cosine > 0.999 on the first half."""
    one = _run_case(case, 1, tmp_path / "one")
    two = _run_case(case, 2, tmp_path / "two")
    kk = one["act_comp"].shape[0]
    c = O.signed_cosines(two["act_comp"].reshape(kk, -1), one["act_comp"].reshape(kk, -1))
    assert c[:k // 2].min() > 0.999, c
    np.testing.assert_allclose(two["act_mean"], one["act_mean"], atol=2e-6)
    np.testing.assert_allclose(two["act_stdev"][:k // 2], one["act_stdev"][:k // 2], rtol=2e-3)
    np.testing.assert_allclose(two["random_stdevs"], one["random_stdevs"], rtol=1e-4)


def test_bench_two_ranks_smoke(tmp_path):
    """``bench.py --gpus 2`` (driver contract: one JSON line from rank 0, whole-job value) with both ranks on the one
    GPU; the two-rank components equal the one-rank components of the same z stream."""
    env = {"GS_BENCH_BACKEND": "gloo", "GS_BENCH_ONE_DEVICE": "1", "GS_BENCH_DUMP": str(tmp_path / "two.npy")}
    outs = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], 2, env)
    line = json.loads([l for l in outs[0][1].splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - 2 * 2 * 50_000 / (line["ms_per_step"] * 2e-3)) < 1e-3 * line["value"]
    assert "cpu_baseline" not in line                      # rank 0 at N = 1 only
    env1 = {"GS_BENCH_DUMP": str(tmp_path / "one.npy")}
    outs1 = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-extras"], 1,
                    dict(env1, GS_BENCH_BACKEND="gloo"))
    line1 = json.loads([l for l in outs1[0][1].splitlines() if l.startswith("{")][-1])
    assert line1["n_gpus"] == 1
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert O.signed_cosines(a[:20], b[:20]).min() > 1 - 1e-6


def test_bench_launches_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` with NO launcher around it (how the driver invokes the N = 1 line; round-5 verdict: the
    same form with N > 1 died on ``assert args.gpus == world``): bench.py re-executes itself under
    ``torch.distributed.run --nproc-per-node 2`` and rank 0 prints the one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GS_BENCH_BACKEND="gloo", GS_BENCH_ONE_DEVICE="1", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"--- stdout ---\n{r.stdout[-3000:]}\n--- stderr ---\n{r.stderr[-6000:]}"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    assert len(line["multi_gpu"]["per_rank"]) == 2

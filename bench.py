#!/usr/bin/env python3
"""Headline benchmark: latent samples/s into the PCA (BASELINE.json metric).

Workload = BASELINE config 2: StyleGAN2-ffhq W-space, -n=1_000_000 -b=10_000 -c=80 on one MI355X with a
random-init mapping network, fed by the reference's z stream (per-batch-seeded NumPy MT19937,
models/wrappers.py:167-174, seed list drawn after np.random.seed(1), decomposition.py:226-227) through the HIP
mapping-network kernel - i.e. through the product's own ``decomposition._presample``.

A *step* is FIVE IPCA blocks (5 x NB = 50 000 rows x 512 features, float32, already resident in HBM) pushed through
``fit_partial`` - in exact mode as ONE call on the contiguous 50 000-row view, exactly as the product's
``decomposition._fit_blocks`` feeds a W-space ``ipca-exact`` fit (the additive statistics do not depend on the
block boundaries); in faithful mode as five NB-row calls (sklearn's recurrence does).  The driver's ``--steps 20``
is therefore exactly n = 1e6 samples per GPU.  After the K timed steps
the job is completed inside the timed region (multi-GPU: the RCCL all-reduce of the sufficient statistics; then
the eigensolve and the device->host copy of the components), so ``value`` is whole-job throughput and
``ms_per_step * steps`` is the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode exact|faithful]     (N > 1: re-executes itself under the launcher below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

With N ranks the job is the 8-GPU config in miniature (BASELINE cfg4 layout): ONE z stream for n = N x K x 50 000
samples, rank r owns the contiguous block range ``distributed.shard_range`` gives it and generates only those z
batches (weak scaling: the per-GPU work is fixed).  Rank 0 prints ONE JSON line (plus notes on stderr).
"""
import argparse
import contextlib
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

D, NB, K_COMP = 512, 10_000, 80
BLOCKS_PER_STEP = 5
RESIDENT_ROWS = None              # set by make_blocks: this rank's W-space rows as one contiguous view
MODEL_SETUP_S = None              # set by make_blocks: construction of the random-init generator (not part of T_sample)
LAUNCH_ROWS = 131072              # rows per Gram launch when the estimator may merge resident rows (gs_ipca_update_resident)
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0             # HBM3E spec
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA peak (the 2:1-sparsity headline figure is not used)
T_BENCH0 = time.perf_counter()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_blocks(n_blocks, dev, rank=0, world=1):
    """The W-space rows the "Fitting batches" loop of cfg2 reads (decomposition.py:245-267), for this rank's share of
    an n = world x n_blocks x NB job: product path end to end (seed protocol -> parallel z generation -> pinned
    H2D -> PixelNorm + 8-layer mapping kernel), resident in HBM.  Returns (blocks, step views, seconds of the
    pre-sampling phase = decomposition.py:226-236, model); building the model is timed apart (MODEL_SETUP_S)."""
    from ganspace_amd import decomposition as dec
    from ganspace_amd.wrappers import get_model
    global MODEL_SETUP_S
    t0 = time.perf_counter()
    model = get_model("StyleGAN2", "ffhq", dev)
    model.use_w()
    torch.cuda.synchronize()
    MODEL_SETUP_S = time.perf_counter() - t0      # (the reference builds its model before the phase timed below)
    t0 = time.perf_counter()
    plan = dec._Plan.make(world * n_blocks * NB, NB, K_COMP)
    assert plan.NB == NB and len(list(plan.block_starts)) == world * n_blocks
    starts = plan.shard_blocks(rank, world)
    b_lo, b_hi = plan.batch_span(starts)
    latent_shape = model.get_latent_shape()          # (draws from the global stream: before the seeding, as in compute())
    torch.manual_seed(dec.SEED_SAMPLING)
    np.random.seed(dec.SEED_SAMPLING)
    latents, row0 = dec._presample(model, plan, latent_shape, dev, b_lo, b_hi)
    torch.cuda.synchronize()
    flat = latents.reshape(latents.shape[0], -1)
    blocks = [flat[g - row0:g - row0 + NB] for g in starts]
    assert all(b - a == NB for a, b in zip(starts[:-1], starts[1:]))
    # one view per step (BLOCKS_PER_STEP consecutive blocks: contiguous rows of the latent array)
    steps = [flat[starts[i] - row0:starts[i] - row0 + BLOCKS_PER_STEP * NB] for i in range(0, len(starts), BLOCKS_PER_STEP)]
    global RESIDENT_ROWS
    RESIDENT_ROWS = flat[starts[0] - row0:starts[-1] - row0 + NB]     # this rank's rows, contiguous
    return blocks, steps, time.perf_counter() - t0, model


def gram_kernel_us(lib, _lib, est, block, iters=50, reps=3):
    """Average duration of the launch the estimator would issue for ``block`` (HIP events on its stream, ``iters``
    back-to-back launches), repeated ``reps`` times: the LAST repetition is returned (the device has been busy for
    tens of ms by then - after host-side phases the first repetition runs 5-12 % slower while the clocks ramp up);
    ``gram_kernel_us.first`` keeps the first repetition for the record."""
    ms, rows = C.c_float(0), C.c_int64(0)
    seen = []
    for _ in range(reps):
        _lib.check(lib.gs_gram_kernel_time(est.transformer._h, C.c_void_p(block.data_ptr()), block.shape[0],
                                           block.stride(0), iters, C.cast(C.byref(ms), C.c_void_p),
                                           C.cast(C.byref(rows), C.c_void_p), _lib.current_stream_ptr()))
        seen.append(ms.value * 1e3)
    gram_kernel_us.first = seen[0]
    return seen[-1], rows.value


def e2e_runs(dev, args, only=None):
    """``get_or_compute`` for BASELINE config 4 at its own n on ONE GPU (StyleGAN2-car Z-space ``--layer=style``, n = 8e6,
    -b 10 000: the N = 1 anchor of the 8-GPU scaling curve, and the one config whose fit loop contains the mapping GEMMs - 800
    blocks x 8 layers - next to a 16 GB resident latent array), for ``ipca-exact`` (the sharded design's estimator) and ``ipca``
    (the reference's default), each checked at a reduced n against scikit-learn on the very blocks the estimator received;
    then for BASELINE config 3 (BigGAN-512 ``generator.gen_z``, n = 1e6, ``-b`` pinned to 2000: the README
    command leaves it to the auto-tuner) and config 5 (StyleGAN2 ``convs.2``, d = 131 072, at the n the run
    run is asked for - BASELINE's 1e6 by default since round 5: the synthetic generator's convolutions go through the f32-MFMA
    GEMM, 24 us per sample).  Phase times from ``decomposition.LAST_TIMINGS``.  cfg3's activation is
    affine in z, so the exact PCA of ALL n activations follows from the 128 x 128 latent covariance: an independent
    float64 reference at the full n for the top components (parity with scikit-learn itself at reduced n:
    tests/test_gpu_decomposition.py)."""
    import contextlib
    import shutil
    import tempfile
    from types import SimpleNamespace
    from ganspace_amd import decomposition as dec
    from ganspace_amd.config import Config
    from ganspace_amd.wrappers import get_instrumented_model
    res = {}
    cfg4 = dict(model="StyleGAN2", layer="style", output_class="car", n=args.e2e_cfg4_n, batch_size=10_000,
                components=K_COMP)
    jobs = (("cfg4_stylegan2_car_zspace_ipca_exact", dict(cfg4, estimator="ipca-exact")),
            ("cfg4_stylegan2_car_zspace_ipca", dict(cfg4, estimator="ipca")),
            ("cfg3_biggan512_gen_z", dict(model="BigGAN-512", layer="generator.gen_z", output_class=250, n=1_000_000,
                                          batch_size=2000, components=K_COMP, estimator="ipca")),
            ("cfg5_stylegan2_convs2", dict(model="StyleGAN2", layer="convs.2", output_class="ffhq", n=args.e2e_cfg5_n,
                                           batch_size=500, components=K_COMP, estimator="ipca")))
    jobs = tuple(j for j in jobs if not only or any(j[0].startswith(o) for o in only))
    for name, kw in jobs:
        run_dir = tempfile.mkdtemp(prefix="gs_bench_e2e_")
        try:
            cfg = Config(**kw)
            if name.startswith("cfg4"):
                # the job's first act is a 16.4 GB device allocation (the resident latents): hand the caching allocator a block
                # of that size first, so that T_sample times the phase and not the driver's one-off VRAM mapping / clearing
                # (0.6 s the first time in this process, 0.00 s from then on; the reference pre-samples into host memory)
                warm = torch.empty(((cfg.n // cfg.batch_size + 2) * cfg.batch_size, 512), dtype=torch.float32, device=dev)
                del warm
            t0 = time.perf_counter()
            inst = get_instrumented_model(cfg.model, cfg.output_class, cfg.layer, dev)
            torch.cuda.synchronize()
            t_model = time.perf_counter() - t0
            dec.PROFILE = True
            with contextlib.redirect_stdout(sys.stderr):
                path = dec.get_or_compute(cfg, inst, submit_config=SimpleNamespace(run_dir_root=run_dir, run_dir=run_dir))
            dec.PROFILE = False
            t = {k_: (round(v, 4) if isinstance(v, float) else v) for k_, v in dec.LAST_TIMINGS.items()}
            data = np.load(path, allow_pickle=False)
            entry = {"config": {k_: v for k_, v in kw.items()}, "model_setup_s": round(t_model, 3), "phases": t,
                     "samples_per_s_total": round(t["n"] / t["T_total_s"], 1),
                     "samples_per_s_fit_loop": round(t["n"] / t["T_fit_loop_s"], 1),
                     "npz_keys_ok": sorted(data.files) == sorted(["act_comp", "act_mean", "act_stdev", "lat_comp", "lat_mean",
                                                                  "lat_stdev", "var_ratio", "random_stdevs"]),
                     "all_finite": bool(all(np.isfinite(data[f]).all() for f in data.files))}
            if name.startswith("cfg3"):
                # activation = Wz z + const (class fixed): covariance = Wz Cov(z) Wz^T; its leading eigenvectors from the
                # 128 x 128 side in float64.  z = the very latents the run used (same seed protocol, regenerated)
                model = inst.model
                g = model.model.generator.gen_z
                Wz = g.weight.detach().double()[:, :128]                       # cond = cat(z, embedding)
                plan = dec._Plan.make(cfg.n, cfg.batch_size, cfg.components)
                lshape = model.get_latent_shape()          # (draws from the global stream: before the seeding, as in compute())
                torch.manual_seed(dec.SEED_SAMPLING)
                np.random.seed(dec.SEED_SAMPLING)
                lat, _ = dec._presample(model, plan, lshape, dev)
                zz = lat.reshape(lat.shape[0], -1)[:plan.N].double()
                zc = zz - zz.mean(0)
                Sz = (zc.T @ zc) / (plan.N - 1)
                Lz = torch.linalg.cholesky(Sz)
                Mz = Wz @ Lz                                                    # [d, 128]: covariance = Mz Mz^T
                w, U = torch.linalg.eigh(Mz.T @ Mz)
                top = torch.argsort(w, descending=True)[:cfg.components]
                comp_ref = (Mz @ U[:, top] / torch.sqrt(w[top])).T.cpu().numpy()
                # every component lies in the 128-dimensional range of Wz; the synthetic gen_z damps its noise columns by
                # 1.05^-j (wrappers._BigGANGenerator), so the leading directions are identifiable: sign-normalised cosine
                # (sklearn's svd_flip rule applied to the reference rows) of the run's act_comp against the exact PCA
                Qw, _ = torch.linalg.qr(Wz)                                      # [d, 128] orthonormal basis of the range
                comp = torch.from_numpy(data["act_comp"].reshape(cfg.components, -1)).to(dev).double()
                inside = torch.linalg.norm(comp @ Qw, dim=1).cpu().numpy()
                ev_ref = np.sqrt(w[top].cpu().numpy())
                ref = torch.from_numpy(comp_ref).to(dev)
                sgn = torch.sign(ref[torch.arange(ref.shape[0]), ref.abs().argmax(dim=1)])
                cosv = ((ref * sgn[:, None]) * comp).sum(1).cpu().numpy()
                entry["vs_exact_pca_of_all_n_activations"] = {
                    "how": "activation is affine in z: exact PCA of all n activations = eigenpairs of Wz Cov(z) Wz^T, from the "
                           "128 x 128 side in float64 (z regenerated with the run's seed protocol)",
                    "top20_signed_cosine_min": round(float(cosv[:20].min()), 7),
                    "top20_signed_cosine_ok": bool(cosv[:20].min() >= 0.999),
                    "all80_abs_cosine_min": round(float(np.abs(cosv).min()), 6),
                    "components_norm_inside_range_of_Wz_min": round(float(inside.min()), 9),
                    "stdev_top20_max_rel_err": float(np.abs(data["act_stdev"][:20] / ev_ref[:20] - 1).max()),
                    "spectrum_sigma1_over_sigma80": round(float(ev_ref[0] / ev_ref[-1]), 3),
                    "note": "IPCA truncates to rank 80 after every 2000-sample block, the reference here is the exact PCA "
                            "(scikit-learn's arithmetic on the same blocks: tests/test_gpu_decomposition.py::"
                            "test_cfg3_biggan_gen_z_small_side)"}
                del comp_ref
                del lat, zz, zc
            if name.startswith("cfg4"):
                # parity at a reduced n ON BOTH SIDES (scikit-learn needs 0.25 s per 10 000-row block: 200 s for the 800
                # blocks): the same job at n = 200 000, every row group the product's estimator receives is also cut into the
                # job's NB-row blocks and handed to sklearn's IncrementalPCA configured as the reference does (estimators.py:59)
                from oracle import reference_cpu
                from oracle.ipca import signed_cosines
                from ganspace_amd import estimators as est_mod
                seen = []
                orig_fit_partial = est_mod.IPCAEstimator.fit_partial

                def spy4(self, X, *a, **k_):
                    seen.append(X.detach().cpu().numpy().copy())
                    return orig_fit_partial(self, X, *a, **k_)
                est_mod.IPCAEstimator.fit_partial = spy4
                try:
                    small = Config(**{**kw, "n": args.e2e_cfg4_check_n})
                    with contextlib.redirect_stdout(sys.stderr):
                        path2 = dec.get_or_compute(small, inst, submit_config=SimpleNamespace(run_dir_root=run_dir, run_dir=run_dir))
                finally:
                    est_mod.IPCAEstimator.fit_partial = orig_fit_partial
                d2 = np.load(path2, allow_pickle=False)
                rows4 = np.concatenate(seen)
                del seen
                nb4 = dec._Plan.make(small.n, small.batch_size, small.components).NB
                sk = reference_cpu.make_reference_ipca(cfg.components)
                with reference_cpu.blas_threads(16):
                    for lo4 in range(0, rows4.shape[0], nb4):
                        sk.partial_fit(rows4[lo4:lo4 + nb4])
                c = signed_cosines(d2["act_comp"].reshape(cfg.components, -1), sk.components_)
                entry["vs_sklearn_at_reduced_n"] = {
                    "n": int(rows4.shape[0]), "blocks": int(rows4.shape[0] // nb4),
                    "top20_signed_cosine_min": round(float(c[:20].min()), 8),
                    "top20_signed_cosine_ok": bool(c[:20].min() >= 0.999),
                    "all80_signed_cosine_min": round(float(c.min()), 8),
                    "all80_abs_cosine_min": round(float(np.abs(c).min()), 6),
                    "act_stdev_top20_max_rel_err": float(np.abs(d2["act_stdev"][:20] /
                                                                np.sqrt(sk.explained_variance_[:20]) - 1).max()),
                    "checker": "scikit-learn IncrementalPCA.partial_fit on the rows the estimator itself received, cut into "
                               "the job's NB-row blocks (ipca-exact: the leading components agree, the trailing ones are the "
                               "exact PCA where IPCA truncates - DESIGN.md 3)"}
                del rows4, sk
            if name.startswith("cfg5"):
                # per-direction parity at a reduced n ON BOTH SIDES (the CPU reference needs 26 s per block at
                # this width): the same job at n = 12 000 (six 2000-row blocks: Rayleigh-Ritz blocks and the deferred-basis
                # blocks of the steady state), every block the product's estimator receives is also handed to the float64
                # restatement of sklearn's recurrence (oracle/smallside_torch.py, pinned to scikit-learn in tests/)
                from oracle.ipca import signed_cosines
                from oracle.smallside_torch import SmallSideTorchOracle
                from ganspace_amd import estimators as est_mod
                orc = SmallSideTorchOracle(cfg.components)
                orig_fit_partial = est_mod.IPCAEstimator.fit_partial

                def spy(self, X, *a, **k_):
                    orc.partial_fit(X)
                    return orig_fit_partial(self, X, *a, **k_)
                est_mod.IPCAEstimator.fit_partial = spy
                try:
                    small = Config(**{**kw, "n": 12_000})
                    with contextlib.redirect_stdout(sys.stderr):
                        path2 = dec.get_or_compute(small, inst, submit_config=SimpleNamespace(run_dir_root=run_dir, run_dir=run_dir))
                finally:
                    est_mod.IPCAEstimator.fit_partial = orig_fit_partial
                d2 = np.load(path2, allow_pickle=False)
                c = signed_cosines(d2["act_comp"].reshape(cfg.components, -1), orc.components_)
                sv = np.sqrt(orc.explained_variance_)
                entry["vs_sklearn_recurrence_at_reduced_n"] = {
                    "n": 12_000, "blocks": int(orc.n_samples_seen_ // 2000),
                    "top20_signed_cosine_min": round(float(c[:20].min()), 8),
                    "top20_signed_cosine_ok": bool(c[:20].min() >= 0.999),
                    "all80_signed_cosine_min": round(float(c.min()), 8),
                    "act_stdev_max_rel_err": float(np.abs(d2["act_stdev"] / sv - 1).max()),
                    "checker": "oracle/smallside_torch.py (float64, the blocks the estimator itself received)"}
                del orc
            res[name] = entry
            inst.close()
            del inst
        except Exception as ex:                                   # an extra must never take the headline line down
            res[name] = {"error": repr(ex)}
            dec.PROFILE = False
        finally:
            shutil.rmtree(run_dir, ignore_errors=True)
            torch.cuda.empty_cache()
    return res


def launcher_command(argv, n_ranks, port=None):
    """The command line that runs this script as ``n_ranks`` ranks on one node - exactly what the driver uses for N > 1:
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...``."""
    if port is None:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__), *argv]


def main():
    global T_BENCH0
    T_BENCH0 = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed steps before the timed region; the default (5 passes over the job, ~25 ms of device "
                         "work) lets the clocks settle: the whole timed job lasts < 5 ms")
    ap.add_argument("--mode", default="exact", choices=["exact", "faithful"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-blocks", type=int, default=16)
    ap.add_argument("--no-wide", action="store_true", help="skip the cfg3/cfg5-shape small-side timings")
    ap.add_argument("--budget-s", type=float, default=330.0,
                    help="wall-clock budget of the whole run: the wide shapes' scikit-learn baselines (26 s per 2000-row block at "
                         "d = 131 072) come last and stop taking blocks when the next one would not fit (>= 2 are always timed)")
    ap.add_argument("--wide-cpu-blocks", type=int, default=6, help="blocks of that baseline (first + steady-state ones)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end get_or_compute runs of cfg3 / cfg5")
    ap.add_argument("--e2e-cfg4-n", type=int, default=8_000_000,
                    help="samples of the cfg4 end-to-end runs (BASELINE's n; 16 GB of resident latents, a few seconds per run)")
    ap.add_argument("--e2e-cfg4-check-n", type=int, default=200_000, help="reduced n of cfg4's scikit-learn comparison")
    ap.add_argument("--e2e-only", default="", help="comma-separated prefixes of end-to-end jobs to run (cfg3, cfg4, cfg5)")
    ap.add_argument("--e2e-cfg5-n", type=int, default=1_000_000,
                    help="samples of the cfg5 end-to-end run (BASELINE's n; ~60 s: the conv prefix runs twice - fit and regression - "
                         "at ~24 us per sample through the f32-MFMA GEMM)")
    ap.add_argument("--wide-cpu-threads", type=int, default=32, help="BLAS threads of that baseline")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (profiling runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` (how the driver's 1-GPU line is invoked, with N > 1): start the N ranks ourselves
        cmd = launcher_command(sys.argv[1:], args.gpus)
        log("bench.py: --gpus %d without a launcher: exec %s" % (args.gpus, " ".join(cmd)))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(cmd[0], cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # test hooks (tests/test_gpu_distributed.py runs two ranks on the ONE GPU of a test box): transport and device
    # placement only - everything measured and computed is the same code
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    if os.environ.get("GS_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from ganspace_amd import _lib, distributed as gdist
    from ganspace_amd.estimators import IPCAEstimator
    lib = _lib.load()

    K, Wm = args.steps, args.warmup
    n_blocks = K * BLOCKS_PER_STEP
    blocks, step_views, t_sample, model = make_blocks(n_blocks, dev, rank, world)
    log(f"rank {rank}: {len(blocks)} blocks of {NB} W-space rows resident ({t_sample:.1f} s: z stream + mapping network)")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ar_events = []

    def run(est, nsteps, finish=True):
        if est.mode == "exact":
            # (the product's W-space loop does the same: decomposition._fit_blocks hands over views of the resident
            #  latent array, the estimator contracts them in launches of 131 072 rows)
            for i in range(nsteps):
                assert est.fit_partial(step_views[i % K], resident=True)
        else:
            for i in range(nsteps * BLOCKS_PER_STEP):
                assert est.fit_partial(blocks[i % n_blocks])
        if finish:
            if dist is not None and args.mode == "exact":
                # (events, no host synchronisation: the exchange's share of the timed region is read afterwards)
                ar_events.clear()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gdist.allreduce_estimator(est, d=D)
                e1.record()
                ar_events.extend((e0, e1))
            est.get_components()           # eigensolve (exact mode) + D2H of the results

    # ---- warm-up (untimed) ------------------------------------------------------------------
    est = IPCAEstimator(K_COMP, args.mode)
    est.transformer._ensure(D)             # allocate the timed handle outside the timed region - and BEFORE the warm-up,
    warm = IPCAEstimator(K_COMP, args.mode)   # so that the device does not idle (and clock down) between the two
    run(warm, Wm, finish=Wm > 0)

    # ---- timed region ------------------------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    run(est, K)
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    samples = K * BLOCKS_PER_STEP * NB * world
    value = samples / dt
    # what a scaling curve needs to explain itself: the exchange step's share of the timed region on every rank, and every
    # rank's pre-sampling time (the ranks of one node share its host cores for the z stream)
    multi = None
    if dist is not None:
        ar_s = ar_events[0].elapsed_time(ar_events[1]) * 1e-3 if len(ar_events) == 2 else 0.0
        mine = {"rank": rank, "allreduce_s": round(ar_s, 6), "T_sample_s": round(t_sample, 4), "timed_region_s": round(dt_local, 6)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        multi = {"per_rank": gathered, "allreduce_s_max": max(g["allreduce_s"] for g in gathered),
                 "T_sample_s_max": max(g["T_sample_s"] for g in gathered),
                 "note": "allreduce_s = HIP events around distributed.allreduce_estimator inside the timed region (header "
                         "all-reduce, re-centring, packed scatter all-reduce, import); T_sample_s is outside it"}
    if rank == 0 and os.environ.get("GS_BENCH_DUMP"):       # test hook: the components the timed job produced
        np.save(os.environ["GS_BENCH_DUMP"], est.get_components()[0])

    # ---- split of the job: update loop vs finalize ----------------------------------------------
    est2 = IPCAEstimator(K_COMP, args.mode)
    est2.transformer._ensure(D)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(est2, K, finish=False)
    torch.cuda.synchronize()
    t_updates = time.perf_counter() - t0
    t0 = time.perf_counter()
    est2.get_components()
    torch.cuda.synchronize()
    t_final = time.perf_counter() - t0
    # ---- the job once more with a pair of HIP events around every Gram compute launch it issues (gs_ipca_profile_launches):
    #      the duration of the dominant kernel AS THE JOB RUNS IT - after a host-side phase, between folds and event waits -
    #      which is what `ms_per_step` contains; the back-to-back microbenchmark below is the steady-state figure
    est3 = IPCAEstimator(K_COMP, args.mode)
    est3.transformer._ensure(D)
    _lib.check(lib.gs_ipca_profile_launches(est3.transformer._h, 1))
    torch.cuda.synchronize()
    run(est3, K, finish=True)
    torch.cuda.synchronize()
    pl_n, pl_ms, pl_rows = C.c_int(0), C.c_double(0.0), C.c_int64(0)
    _lib.check(lib.gs_ipca_launch_profile(est3.transformer._h, C.cast(C.byref(pl_n), C.c_void_p),
                                          C.cast(C.byref(pl_ms), C.c_void_p), C.cast(C.byref(pl_rows), C.c_void_p)))
    _lib.check(lib.gs_ipca_profile_launches(est3.transformer._h, 0))
    injob = None
    if pl_n.value > 0 and pl_ms.value > 0:
        injob = {"launches": pl_n.value, "rows": pl_rows.value, "total_ms": pl_ms.value,
                 "tflops": pl_rows.value * D * (D + 1) / (pl_ms.value * 1e-3) / 1e12,
                 "gbs": pl_rows.value * D * 4 / (pl_ms.value * 1e-3) / 1e9}
    del est3

    # ---- roofline of the dominant kernel of the timed region (the partial X^T X MFMA kernel: one launch per
    #      block, K x 5 launches), HIP events on its stream ----------------------------------------------------
    if args.mode == "exact":
        gram_view = RESIDENT_ROWS[:LAUNCH_ROWS]
        n_launches = -(-K * BLOCKS_PER_STEP * NB // LAUNCH_ROWS)
    else:
        gram_view = blocks[0]
        n_launches = n_blocks
    us, rows_l = gram_kernel_us(lib, _lib, est2, gram_view)
    us_first = gram_kernel_us.first
    flops = rows_l * D * (D + 1)           # algorithmic: upper triangle incl. diagonal, 2 flop/MAC
    bytes_ = rows_l * D * 4                # algorithmic: one read of the [rows, d] f32 block
    ach_tf = flops / (us * 1e-6) / 1e12
    ach_gbs = bytes_ / (us * 1e-6) / 1e9
    # HBM traffic of that kernel from the committed rocprofv3 PMC passes (bench.py cannot run the profiler on
    # itself): FETCH_SIZE x2-corrected as MI355X_MICROARCH.md prescribes + WRITE_SIZE, per launch
    traffic = None
    traffic_note = None
    try:
        with open(os.path.join(ROOT, "profiles", "gram_pmc_latest.json")) as f:
            pmc = json.load(f)
        pmc = pmc.get("f32", pmc)              # (tools/summarize_r03.py writes one section per precision)
        if pmc.get("rows_per_launch") == rows_l:
            traffic = pmc.get("hbm_bytes_per_launch", pmc["hbm_read_bytes_per_launch_corrected_x2"])
            traffic_note = pmc.get("traffic_breakdown") or (
                "%.1f MB read + %.1f MB written per launch against %.1f MB of X rows (algorithmic): the rows are fetched "
                "once; the rest is the float32 partial-Gram slabs (written by this launch, read back by the fold)"
                % (pmc["hbm_read_bytes_per_launch_corrected_x2"] / 1e6, pmc["hbm_write_bytes_per_launch"] / 1e6,
                   bytes_ / 1e6))
    except Exception:
        pass
    frac_of_region = (n_launches * us * 1e-6) / (t_updates + t_final) if (t_updates + t_final) > 0 else None
    # `achieved` / `frac`: the kernel inside the job (events around the job's own launches); the steady-state microbenchmark
    # (50 back-to-back launches, third repetition) is reported next to it, not instead of it
    job_tf = injob["tflops"] if injob else ach_tf
    roofline = {"bound": "mfma", "kernel": "gram_f32_wide_kernel" if args.mode == "exact" else "gram_partial_kernel<true, false>",
                "achieved": round(job_tf, 2),
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(job_tf / PEAK_F32_MFMA_TFLOPS, 4),
                "frac_source": ("HIP events around the %d Gram launches of one run of the timed job (%d rows, %.3f ms in "
                                "total; gs_ipca_profile_launches)" % (injob["launches"], injob["rows"], injob["total_ms"]))
                               if injob else "steady-state microbenchmark (no in-job profile)",
                "in_job_avg_launch_us": None if not injob else round(injob["total_ms"] * 1e3 / injob["launches"], 2),
                "steady_state_microbenchmark": {"achieved": round(ach_tf, 2), "frac": round(ach_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                                "avg_launch_us": round(us, 2), "rows_per_launch": rows_l},
                "traffic": traffic, "traffic_source": "profiles/gram_pmc_latest.json (rocprofv3 --pmc: FETCH_SIZE x2 + WRITE_SIZE)",
                "traffic_note": traffic_note,
                "avg_launch_us": round(us, 2), "avg_launch_us_first_repetition": round(us_first, 2),
                "launch_timing": "HIP events around 50 back-to-back launches on the estimator's stream, third repetition "
                                 "(the first one follows host-side phases and runs while the clocks ramp up)",
                "rows_per_launch": rows_l, "launches_in_timed_region": n_launches,
                "share_of_timed_region": None if frac_of_region is None else round(frac_of_region, 3),
                "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": bytes_,
                "hbm_achieved_GBs": round(injob["gbs"] if injob else ach_gbs, 1),
                "hbm_frac_of_8TBs": round((injob["gbs"] if injob else ach_gbs) / PEAK_HBM_GBS, 4),
                "clock_note": "peak = 2.4 GHz x 256 CU x 4 SIMD x 64 flop/clk; the s_memtime traces of this kernel "
                              "(DESIGN.md 5) show ~2.05 GHz under ITS load (MFMA + LDS + VALU mix) - a property of the "
                              "kernel's power draw, not a chip limit (a pure MFMA loop holds 155 TF, MI355X_MICROARCH.md)"}

    out = {
        "metric": "latent samples/sec into PCA (n=1e6) + top-20 component cos-sim vs reference",
        "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg2: StyleGAN2-ffhq W-space PCA, -n=%d -b=10_000 -c=80 per GPU (%d in total), "
                               "random-init mapping network, reference z stream (MT19937 per-batch seeds), "
                               "activations resident in HBM" % (K * BLOCKS_PER_STEP * NB, samples),
                   "mode": args.mode, "feat_dim": D, "block_rows": NB, "blocks_per_step": BLOCKS_PER_STEP,
                   "components": K_COMP,
                   "parallelism": f"dp{world} (blocks of one z stream sharded, one RCCL all-reduce of n/mean/scatter)"},
        "roofline": roofline,
        "breakdown": {"update_loop_s": round(t_updates, 5), "finalize_eigensolve_s": round(t_final, 5),
                      "sampling_zgen_plus_mapping_s": round(t_sample, 4),
                      "model_setup_s": None if MODEL_SETUP_S is None else round(MODEL_SETUP_S, 4),
                      "eigh_products": int(lib.gs_ipca_last_mults(est2.transformer._h)),
                      "eigh_sweeps": int(lib.gs_ipca_last_sweeps(est2.transformer._h))},
    }
    if multi is not None:
        out["multi_gpu"] = multi

    extras = rank == 0 and world == 1 and not args.no_extras
    # ---- CPU baseline + cos-sim on a bounded sample (rank 0, N=1 only) -----------------------------
    if extras and not args.no_cpu_baseline:
        from oracle import reference_cpu
        from oracle.ipca import signed_cosines
        nb = min(args.cpu_blocks, n_blocks)
        host_blocks = [b.cpu().numpy() for b in blocks[:nb]]
        # BLAS thread count: more threads than the (k + NB + 1) x 512 SVD can feed make LAPACK slower, so a few settings
        # are tried on two steady-state blocks each and the fastest is used (and stated) - the baseline is the best
        # the host does, not an accident of the default
        threads, tried = reference_cpu.pick_blas_threads(host_blocks[:3], K_COMP)
        with reference_cpu.blas_threads(threads):
            ref, per_block = reference_cpu.time_reference_fit(host_blocks, K_COMP, per_block=True)
        t_cpu, n_cpu = float(sum(per_block)), nb * NB
        steady = float(np.mean(per_block[1:])) if nb > 1 else per_block[0]
        # whole cfg2 job on the CPU (100 blocks per 1e6 samples): first block (float32 SVD, sklearn's dtype behaviour)
        # + steady-state blocks, extrapolated linearly (cost per block is constant after block 1)
        job_blocks = K * BLOCKS_PER_STEP
        t_job_cpu = per_block[0] + (job_blocks - 1) * steady
        out["cpu_baseline"] = {"value": round(job_blocks * NB / t_job_cpu, 1), "unit": "samples/s", "cores": threads,
                               "kind": "reference",
                               "sample": f"first {nb} of the {n_blocks} blocks ({n_cpu} samples, {t_cpu:.1f} s) timed: "
                                         "sklearn IncrementalPCA.partial_fit configured as estimators.py:59 (the "
                                         "arithmetic the reference executes); value = the whole job extrapolated "
                                         "from the first block + steady-state blocks",
                               "first_block_s": round(per_block[0], 4), "steady_block_s": round(steady, 4),
                               "steady_samples_per_s": round(NB / steady, 1),
                               "measured_sample_samples_per_s": round(n_cpu / t_cpu, 1),
                               "blas_threads_tried_s_per_block": tried, "host_cpu_count": os.cpu_count()}
        # ---- the TIMED estimator (131 072-row launches of the wide kernel, all K x 5 blocks) against a float64 exact
        #      PCA of the very rows it consumed: plain torch float64 matmuls on the device, LAPACK eigh on the host
        from oracle.ipca import flip_rows_largest_abs_positive
        rows_all = RESIDENT_ROWS[:n_blocks * NB]
        pilot = rows_all[:NB].double().mean(0)
        G = torch.zeros((D, D), dtype=torch.float64, device=dev)
        s1 = torch.zeros(D, dtype=torch.float64, device=dev)
        for lo in range(0, rows_all.shape[0], 100_000):
            c = rows_all[lo:lo + 100_000].double() - pilot
            G += c.T @ c
            s1 += c.sum(0)
        n_all = rows_all.shape[0]
        Cc = (G - torch.outer(s1, s1) / n_all).cpu().numpy()
        w_ref, V_ref = np.linalg.eigh(Cc)
        V_ref = V_ref[:, ::-1][:, :K_COMP].T
        V_ref = V_ref * flip_rows_largest_abs_positive(V_ref)[:, None]
        sv_ref = np.sqrt(np.maximum(w_ref[::-1][:K_COMP], 0.0))
        if args.mode == "exact":
            ct = signed_cosines(est.get_components()[0], V_ref)
            out["timed_estimator_check"] = {
                "against": "float64 exact PCA of the same %d rows (torch float64 on the device + LAPACK eigh)" % n_all,
                "top20_min_signed_cos": round(float(ct[:20].min()), 9), "all80_min_signed_cos": round(float(ct.min()), 9),
                "singular_values_max_rel_err": float(np.abs(est.transformer.singular_values_ / sv_ref - 1).max()),
                "mean_max_abs_err": float(np.abs(est.transformer.mean_ - (pilot + s1 / n_all).cpu().numpy()).max())}
        # ---- T_sample / T_total: the reference's pre-sampling phase (decomposition.py:232-236: one
        #      sample_latent per batch = per-batch-seeded MT19937 normals on one core, models/wrappers.py:167-174, then
        #      the mapping network) timed on the host for two batches and extrapolated to the job's n_lat // B batches
        from oracle import synth
        Wm_, bm_ = (model.model.style.weight.detach().cpu().numpy(), model.model.style.bias.detach().cpu().numpy())
        n_batches_job = (job_blocks * NB + NB - 1) // NB + 1
        t_z, t_map = [], []
        for seed in (1791095845, 2135392491):
            t0 = time.perf_counter()
            z = np.random.RandomState(seed).standard_normal(512 * NB).reshape(NB, 512).astype(np.float32)
            t_z.append(time.perf_counter() - t0)
            with reference_cpu.blas_threads(threads):
                t0 = time.perf_counter()
                synth.mapping_network(z, Wm_, bm_, lr_mul=0.01, dtype=np.float32)
                t_map.append(time.perf_counter() - t0)
        cpu_sample = n_batches_job * (min(t_z) + min(t_map))
        gpu_total = t_sample + dt
        out["end_to_end"] = {
            "gpu": {"T_sample_s": round(t_sample, 3), "T_fit_s": round(dt, 5), "T_total_s": round(gpu_total, 3),
                    "samples_per_s": round(samples / gpu_total, 1),
                    "note": "T_sample = z stream from the device generator (gs_zgen_device) + HIP mapping network; T_fit = the "
                            "timed region of `value`"},
            "cpu": {"T_sample_s": round(cpu_sample, 2), "T_fit_s": round(t_job_cpu, 2),
                    "T_total_s": round(cpu_sample + t_job_cpu, 2),
                    "samples_per_s": round(job_blocks * NB / (cpu_sample + t_job_cpu), 1),
                    "z_s_per_batch": round(min(t_z), 4), "mapping_s_per_batch": round(min(t_map), 4),
                    "kind": "port (NumPy float32 mapping network of oracle/synth.py; the reference runs it on its "
                            "torch device) + the reference's own RNG calls",
                    "sample": f"2 of the {n_batches_job} batches of {NB} latents, extrapolated"},
            "speedup_total": round((cpu_sample + t_job_cpu) / gpu_total, 1)}
        # ---- regression back to latent space, timed separately (decomposition.py:77-139):
        #      the Z-space form of the same job (layer "style": the activation of z IS the W row that was fitted), one
        #      rank's share of cfg4 = 10^6 fresh samples in mini-batches of 10 000.  Device: native z stream -> mapping
        #      network -> projection -> [A|Z]^T [A|Z] by the Gram kernel -> k x k solve.  Host: the reference's steps on a
        #      sample (RNG + mapping per batch as timed above, gelsd on 100 000 rows, both linear in n) --------------
        try:
            import scipy.linalg
            from types import SimpleNamespace
            from ganspace_amd import decomposition as dec
            from ganspace_amd.nethook import InstrumentedModel
            comp_t, stdev_t, _ = est.get_components()
            n_reg = job_blocks * NB
            model.use_z()
            inst_r = InstrumentedModel(model)
            inst_r.retain_layer("style")
            saved_b, dec.B = dec.B, NB
            reg_runs = []
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(sys.stderr):
                    z_comp, z_mean = dec.linreg_lstsq(comp_t, est.transformer.mean_, stdev_t, inst_r,
                                                      SimpleNamespace(n=n_reg, layer="style"))
                torch.cuda.synchronize()
                reg_runs.append(time.perf_counter() - t0)
            dec.B = saved_b
            inst_r.close()
            model.use_w()
            rs = np.random.RandomState(3)
            rows_s = 100_000
            A_s = rs.standard_normal((rows_s, K_COMP)).astype(np.float32)
            Z_s = (A_s @ rs.standard_normal((K_COMP, 512)).astype(np.float32)
                   + rs.standard_normal((rows_s, 512)).astype(np.float32))
            with reference_cpu.blas_threads(threads):
                t0 = time.perf_counter()
                scipy.linalg.lstsq(A_s, Z_s, lapack_driver="gelsd")
                t_gelsd = time.perf_counter() - t0
            cpu_reg = n_reg / NB * (min(t_z) + min(t_map)) + t_gelsd * n_reg / rows_s
            out["regression_cfg4_share"] = {
                "samples": n_reg, "mini_batch": NB, "latent_dims": 512, "components": K_COMP,
                "gpu_s": round(min(reg_runs), 3), "gpu_s_runs": [round(r, 3) for r in reg_runs],
                "gpu_samples_per_s": round(n_reg / min(reg_runs), 1),
                "solution_finite": bool(np.isfinite(z_comp).all() and np.isfinite(z_mean).all()),
                "cpu_s_extrapolated": round(cpu_reg, 1),
                "cpu_parts": {"z_s_per_batch": round(min(t_z), 4), "mapping_s_per_batch": round(min(t_map), 4),
                              "gelsd_s_per_100k_rows": round(t_gelsd, 3), "blas_threads": threads},
                "speedup": round(cpu_reg / min(reg_runs), 1),
                "note": "z streams from the device generator (gs_zgen_device), mapping network, projection and "
                        "[A|Z]^T [A|Z] all on the device; 8 mini-batches per partial_forward call (decomposition._forward_rows)"}
        except Exception as ex:                                   # an extra must never take the headline line down
            out["regression_cfg4_share"] = {"error": repr(ex)}
        cos = {}
        for mode in ("exact", "faithful"):
            e = IPCAEstimator(K_COMP, mode)
            for b in blocks[:nb]:
                e.fit_partial(b)
            c = signed_cosines(e.get_components()[0], ref.components_)
            cos[mode] = {"top20_min_signed_cos": round(float(c[:20].min()), 7),
                         "all80_min_abs_cos": round(float(np.abs(c).min()), 5)}
            if mode == "faithful":
                # throughput of the sklearn-faithful mode on the same sample (one eigensolve per block)
                torch.cuda.synchronize()
                e2 = IPCAEstimator(K_COMP, mode)
                t0 = time.perf_counter()
                for b in blocks[:nb]:
                    e2.fit_partial(b)
                e2.get_components()
                torch.cuda.synchronize()
                cos[mode]["samples_per_s"] = round(nb * NB / (time.perf_counter() - t0), 1)
        cos["sample"] = (f"the first {nb} blocks ({nb * NB} rows) on both sides: scikit-learn on the host is the slow side; the "
                         "n = 1e6 comparison with scikit-learn runs in tests/test_gpu_decomposition.py (cfg2_full)")
        out["cos_sim_vs_reference"] = cos
        out["vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)

        # ---- the same job in the sklearn-faithful mode (the reference's own recurrence: truncation to k components
        #      after every block), whole n ------------------------------------------------------------------------
        if args.mode == "exact":
            ef = IPCAEstimator(K_COMP, "faithful")
            for b in blocks[:BLOCKS_PER_STEP * max(1, Wm)]:
                ef.fit_partial(b)
            ef.get_components()
            runs = []
            for rep in range(3):           # one host round trip per block: sensitive to what else the host is doing
                ef = IPCAEstimator(K_COMP, "faithful")
                ef.transformer._ensure(D)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for b in blocks:
                    ef.fit_partial(b)
                ef.get_components()
                torch.cuda.synchronize()
                runs.append(time.perf_counter() - t0)
            tf_ = min(runs)
            out["faithful_mode_same_job"] = {"samples_per_s": round(n_blocks * NB / tf_, 1),
                                             "ms_per_block": round(tf_ / n_blocks * 1e3, 4), "blocks": n_blocks,
                                             "job_s_of_3_runs": [round(r, 5) for r in runs]}

        # ---- opt-in bf16 contraction modes (precision="bf16x6" / "bf16x3": split, float32-class / 2^-16; "bf16": one
        #      plane, the HBM-bound single-pass contraction of SURVEY.md 8d; the headline stays exact f32):
        #      same job, same timed region; cos-sim against the same sklearn reference sample ------------------
        if args.mode == "exact":
            split = {}
            for prec, nprod in (("bf16x6", 6), ("bf16x3", 3), ("bf16", 1)):
                e = IPCAEstimator(K_COMP, "exact", precision=prec)
                for b in blocks[:nb]:
                    e.fit_partial(b)
                c = signed_cosines(e.get_components()[0], ref.components_)
                e2 = IPCAEstimator(K_COMP, "exact", precision=prec)
                e2.transformer._ensure(D)
                run(e2, Wm)
                e2 = IPCAEstimator(K_COMP, "exact", precision=prec)
                e2.transformer._ensure(D)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(e2, K)
                torch.cuda.synchronize()
                job = time.perf_counter() - t0
                # the launch the resident path issues: one "wide" launch (pairs of workgroups hold the whole upper
                # triangle) of 131 072 rows for bf16x3 and of up to 2^20 rows for bf16 (8192-row chunks: all the
                # resident rows of this job); bf16x6 launches are capped at 24 576 rows (no float64 carry)
                us_p, rt = gram_kernel_us(lib, _lib, e2, RESIDENT_ROWS[:LAUNCH_ROWS * (8 if prec == "bf16" else 1)])
                us_randn = None
                if prec == "bf16":
                    # the same launch on N(0,1) rows of the same shape: the kernel's duration depends on the DATA
                    # (bit toggling -> power -> clock), so both are reported; `roofline_hbm` prices the job's own rows
                    Xr = torch.randn(rt, D, device=dev)
                    us_randn, _ = gram_kernel_us(lib, _lib, e2, Xr)
                    del Xr
                mfma_tf = nprod * rt * D * (D + 1) / (us_p * 1e-6) / 1e12
                gbs = rt * D * 4 / (us_p * 1e-6) / 1e9
                split[prec] = {"samples_per_s": round(n_blocks * NB / job, 1), "gram_launch_us": round(us_p, 2),
                               "rows_per_launch": rt,
                               **({"gram_launch_us_on_randn_rows": round(us_randn, 2),
                                   "hbm_frac_on_randn_rows": round(rt * D * 4 / (us_randn * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)}
                                  if us_randn else {}),
                               "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                                            "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)},
                               "bf16_mfma_TFLOPs_executed": round(mfma_tf, 1),
                               "frac_of_bf16_dense_peak": round(mfma_tf / PEAK_BF16_MFMA_TFLOPS, 4),
                               "top20_min_signed_cos": round(float(c[:20].min()), 7),
                               "all80_min_abs_cos": round(float(np.abs(c).min()), 5)}
            out["split_bf16_modes"] = split
            # the HBM-bound contraction ("bf16 single-pass") as a second roofline entry: opt-in
            # (precision="bf16"), leading components inside the north_star tolerance - see its cos fields
            b = split["bf16"]
            hb_traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", "gram_pmc_latest.json")) as f:
                    pm = json.load(f).get("bf16", {})
                if pm.get("rows_per_launch") == b["rows_per_launch"]:
                    hb_traffic = pm.get("hbm_bytes_per_launch")
            except Exception:
                pass
            out["roofline_hbm"] = {"bound": "hbm", "kernel": "gram_bf16_glds_kernel (precision=\"bf16\", opt-in)",
                                   "achieved": b["roofline"]["achieved"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": b["roofline"]["frac"], "traffic": hb_traffic,
                                   "avg_launch_us": b["gram_launch_us"], "rows_per_launch": b["rows_per_launch"],
                                   "avg_launch_us_on_randn_rows": b.get("gram_launch_us_on_randn_rows"),
                                   "frac_on_randn_rows": b.get("hbm_frac_on_randn_rows"),
                                   "algorithmic_bytes_per_launch": b["rows_per_launch"] * D * 4,
                                   "samples_per_s_whole_job": b["samples_per_s"],
                                   "top20_min_signed_cos_vs_sklearn": b["top20_min_signed_cos"],
                                   "read_ceiling_note": "a streaming-read kernel with this access pattern and no arithmetic "
                                                        "reaches 5.45 TB/s on this chip (tools/ubench/read_bw.hip, "
                                                        "profiles/r03_probes.md)"}

    # ---- the wide-feature BASELINE shapes (cfg3 d = 32 768, cfg5 d = 131 072; NB = 2 000, k = 80): PCA-only
    #      throughput of the small-side recurrence on the synthetic low-rank-plus-noise blocks of SURVEY.md 8d item 5,
    #      checked in the same run against a float64 restatement of the recurrence (oracle/smallside_torch.py) and timed
    #      next to scikit-learn on the host ------------------------------------------------------------------------
    if extras and not args.no_wide:
        from oracle import reference_cpu
        from oracle.ipca import signed_cosines
        from oracle.smallside_torch import SmallSideTorchOracle, lowrank_plus_noise_blocks
        wide = {}
        n_wide = 12
        for name, dd in (("cfg3_shape_d32768", 32768), ("cfg5_shape_d131072", 131072)):
            entry = {"block_rows": 2000, "mode": "ipca (small-side, sklearn-faithful)", "blocks": n_wide}
            ests = {p_: IPCAEstimator(K_COMP, "faithful", precision=p_) for p_ in ("f32", "bf16x6")}
            ts = {p_: [] for p_ in ests}
            # pass 1: timing only (a host-side step between the blocks - the oracle's LAPACK eigh takes a second -
            # lets the GPU drop its clocks and the next block pays the wake-up)
            for p_, e in ests.items():
                for X in lowrank_plus_noise_blocks(dd, n_wide, rows=2000, device=dev):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    e.fit_partial(X)
                    torch.cuda.synchronize()
                    ts[p_].append(time.perf_counter() - t0)
                    del X
            # pass 2: the float64 oracle on the same (regenerated) blocks
            orc = SmallSideTorchOracle(K_COMP)
            for i, X in enumerate(lowrank_plus_noise_blocks(dd, n_wide, rows=2000, device=dev)):
                orc.partial_fit(X)
                del X
            r_ = K_COMP + 2000 + 1
            # executed work of a steady-state block, triangle convention (as the headline's roofline): T = M M^T over its
            # upper triangle r (r + 1) d, W = Q^T M 2 k r d;  SURVEY.md 8d's per-sample figure 2 d (m + 2k) counts
            # the full products X X^T and V X^T (listed separately, not comparable with the headline's convention)
            flops_tri = float(r_) * (r_ + 1) * dd + 2.0 * K_COMP * r_ * dd
            flops_survey = 2000 * 2.0 * dd * (2000 + 2 * K_COMP)
            for p_, e in ests.items():
                # blocks 2-4: Rayleigh-Ritz solve per block; from the fifth block on the recurrence carries an
                # undiagonalised basis (steady state of a long fit)
                early = sum(ts[p_][1:4]) / 3
                steady = sum(ts[p_][6:]) / len(ts[p_][6:])
                c = signed_cosines(e.get_components()[0], orc.components_)
                peak = PEAK_F32_MFMA_TFLOPS if p_ == "f32" else PEAK_BF16_MFMA_TFLOPS / 6.0
                entry[p_] = {"ms_per_block": round(steady * 1e3, 2), "samples_per_s": round(2000 / steady, 1),
                             "executed_TFLOPs_triangle_convention": round(flops_tri / steady / 1e12, 1),
                             "frac_of_peak_whole_block": round(flops_tri / steady / 1e12 / peak, 3),
                             "survey_convention_full_product_TFLOPs": round(flops_survey / steady / 1e12, 1),
                             "ms_per_block_first_blocks": round(early * 1e3, 2),
                             "vs_float64_oracle_all80_min_signed_cos": round(float(c.min()), 8),
                             "vs_float64_oracle_top20_min_signed_cos": round(float(c[:20].min()), 8),
                             "singular_values_max_rel_err": float(np.abs(e.transformer.singular_values_ /
                                                                         orc.singular_values_ - 1).max())}
            del ests, orc
            entry["ms_per_block"] = entry["f32"]["ms_per_block"]
            entry["samples_per_s"] = entry["f32"]["samples_per_s"]
            wide[name] = entry
            torch.cuda.empty_cache()
        out["wide_feature_shapes"] = wide

    # ---- end to end, BASELINE configs 3 and 5 through the product's own get_or_compute (decomposition.py:226-341 of the
    #      reference: pre-sampling, "Fitting batches" loop with the hooked generator, read-out, regression back to latent
    #      space, .npz), phase by phase (ganspace_amd.decomposition.LAST_TIMINGS) ---------------------------------------
    if extras and not args.no_e2e:
        out["end_to_end_cfg3_cfg4_cfg5"] = e2e_runs(dev, args, [o for o in args.e2e_only.split(",") if o])

    # ---- the wide shapes' CPU baselines, LAST and alone on the host (beside the GPU legs they ran 2-5x slower and slowed
    #      the host-bound GPU legs down in turn): scikit-learn's IncrementalPCA.partial_fit - the reference's arithmetic,
    #      /root/reference/estimators.py:68-76 - on CPU-seeded blocks of the same synthetic workload, first block + up to
    #      `--wide-cpu-blocks - 1` steady-state blocks (SURVEY.md 8d asks for >= 5) while the run's budget lasts; the n = 1e6
    #      job (500 blocks) is extrapolated.  The same blocks then go through the device estimator for the cosine ----------
    if extras and not args.no_wide and not args.no_cpu_baseline:
        from oracle import reference_cpu
        from oracle.ipca import signed_cosines
        from oracle.smallside_torch import lowrank_plus_noise_blocks
        for name, dd in (("cfg3_shape_d32768", 32768), ("cfg5_shape_d131072", 131072)):
            entry = out.get("wide_feature_shapes", {}).get(name)
            if entry is None:
                continue
            sk = reference_cpu.make_reference_ipca(K_COMP)
            per = []
            with reference_cpu.blas_threads(args.wide_cpu_threads):
                for X in lowrank_plus_noise_blocks(dd, args.wide_cpu_blocks, rows=2000, device="cpu"):
                    guess = per[-1] if len(per) >= 2 else (per[0] if per else 0.0)
                    if len(per) >= 2 and (time.perf_counter() - T_BENCH0) + guess > args.budget_s:
                        break
                    hb = X.numpy()
                    t0 = time.perf_counter()
                    sk.partial_fit(hb)
                    per.append(time.perf_counter() - t0)
                    del X, hb
            steady_cpu = float(np.mean(per[1:]))
            t_job = per[0] + 499 * steady_cpu
            entry["cpu_baseline"] = {"value": round(1e6 / t_job, 2), "unit": "samples/s", "kind": "reference",
                                     "cores": args.wide_cpu_threads, "first_block_s": round(per[0], 2),
                                     "steady_block_s": round(steady_cpu, 2), "steady_blocks_timed": len(per) - 1,
                                     "seconds_per_block": [round(x, 2) for x in per],
                                     "sample": f"{len(per)} blocks of 2000 x {dd} (sklearn IncrementalPCA.partial_fit, alone on the "
                                               "host at the end of the run), n = 1e6 (500 blocks) extrapolated"}
            entry["vs_cpu_baseline"] = round(entry["f32"]["samples_per_s"] / entry["cpu_baseline"]["value"], 1)
            e = IPCAEstimator(K_COMP, "faithful")
            for X in lowrank_plus_noise_blocks(dd, len(per), rows=2000, device="cpu"):
                e.fit_partial(X.to(dev))
            c = signed_cosines(e.get_components()[0], sk.components_)
            entry["vs_sklearn_at_reduced_n"] = {"n": len(per) * 2000, "all80_min_signed_cos": round(float(c.min()), 8),
                                                "top20_min_signed_cos": round(float(c[:20].min()), 8)}
            del e, sk

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

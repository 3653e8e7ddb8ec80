"""Component-discovery driver: the reference's ``decomposition.py`` hot loop, device-resident.

Mirrors ``/root/reference/decomposition.py``: ``get_or_compute`` (:362-368) -> ``_compute``
(:370-402, cache-file naming) -> ``compute`` (:150-358).  Same seeds, same batch/block index
arithmetic (N = n // B * B, NB = max(B, 2000, 3k), block gi = latents[gi : gi+NB], tail
mini-batch truncated), same ``.npz`` schema (:331-341) - so ``interactive.py`` /
``visualize.py`` style consumers can load the result unchanged.

What is different (the point of this implementation):
  * the pre-sampled ``latents`` array lives in HBM instead of host RAM (2 GB at n = 1e6) and
    the per-mini-batch H2D (:247) / D2H (:261) copies are gone: the hooked activation is
    written straight into the device block buffer ``X[NB, d]``;
  * ``fit_partial`` receives that device buffer and runs the MFMA Gram update / eigensolver
    (``ganspace_amd.estimators``) instead of sklearn on the host;
  * the latent-space regression (``linreg_lstsq``, :77-139) accumulates the normal equations
    ``[A|Z]^T [A|Z]`` with the same Gram kernel instead of materialising ``A[n,k]``/``Z[n,512]``
    on the host and calling LAPACK gelsd.
"""
from __future__ import annotations

import datetime
import os
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from . import _zgen, ops
from .estimators import get_estimator
from .nethook import InstrumentedModel
from .wrappers import get_instrumented_model

SEED_SAMPLING = 1
SEED_RANDOM_DIRS = 2
SEED_LINREG = 3
SEED_VISUALIZATION = 5

B = 20


def _progress(it, desc=None):
    try:
        from tqdm import tqdm
        return tqdm(it, desc=desc, ascii=True)
    except Exception:
        return it


def get_random_dirs(components, dimensions):
    """decomposition.py:42-46."""
    gen = np.random.RandomState(seed=SEED_RANDOM_DIRS)
    dirs = gen.normal(size=(components, dimensions))
    dirs /= np.sqrt(np.sum(dirs ** 2, axis=1, keepdims=True))
    return dirs.astype(np.float32)


def get_max_batch_size(inst, device, layer_name=None):
    """decomposition.py:49-74: largest even batch <= 20 that keeps peak memory under half the VRAM."""
    inst.remove_edits()
    torch.cuda.reset_peak_memory_stats(device)
    total_mem = torch.cuda.get_device_properties(device).total_memory
    B_max = 20
    for i in range(2, B_max, 2):
        z = inst.model.sample_latent(n_samples=i)
        if layer_name:
            inst.model.partial_forward(z, layer_name)
        else:
            inst.model.forward(z)
        maxmem = torch.cuda.max_memory_allocated(device)
        del z
        if maxmem > 0.5 * total_mem:
            print("Batch size {:d}: memory usage {:.0f}MB".format(i, maxmem / 1e6))
            return i
    return B_max


def linreg_lstsq(comp_np, mean_np, stdev_np, inst, config):
    """Directions in latent space that reproduce the activation-space PCs (decomposition.py:77-139).

    min_M || A M - Z ||  with  A = ((G'(Z) - mean) P^T) / stdev.  Solved through the normal equations:
    both ``A^T A`` (k x k) and ``A^T Z`` (k x latent) are blocks of ``[A|Z]^T [A|Z]``, which the
    MFMA Gram kernel accumulates over mini-batches (float64 across chunks); the k x k solve runs
    on the host in float64.
    """
    print("Performing least squares regression", flush=True)
    torch.manual_seed(SEED_LINREG)
    np.random.seed(SEED_LINREG)
    dev = inst.model.device
    comp = torch.from_numpy(comp_np).float().to(dev)
    mean = torch.from_numpy(np.asarray(mean_np)).float().to(dev).reshape(1, -1)
    stdev = torch.from_numpy(stdev_np).float().to(dev)
    n_samp = max(10_000, config.n) // B * B
    n_comp = comp.shape[0]
    latent_dims = int(inst.model.get_latent_dims())
    dcat = n_comp + latent_dims
    pad = (-dcat) % 4
    G = torch.zeros((dcat + pad, dcat + pad), dtype=torch.float64, device=dev)
    cs = torch.zeros(dcat + pad, dtype=torch.float64, device=dev)
    bias = -(comp.double() @ mean.double().reshape(-1)).float()
    with torch.no_grad():
        for _ in _progress(range(n_samp // B), desc="Collecting samples"):
            z = inst.model.sample_latent(B)
            inst.model.partial_forward(z, config.layer)
            act = inst.retained_features()[config.layer].reshape(B, -1)
            coords = ops.linear_forward(act, comp, bias) if act.shape[1] % 4 == 0 else (act - mean) @ comp.T
            AZ = torch.cat([coords / stdev, z.reshape(B, -1).float(),
                            torch.zeros((B, pad), device=dev)], dim=1).contiguous()
            ops.gram_accumulate(AZ, G, cs)
    Gh = G.cpu().numpy()
    AtA, AtZ = Gh[:n_comp, :n_comp], Gh[:n_comp, n_comp:dcat]
    M_t = np.linalg.lstsq(AtA, AtZ, rcond=None)[0]
    Z_comp = M_t[:n_comp, :]
    Z_mean = (cs.cpu().numpy()[n_comp:dcat] / n_samp)[None, :]
    return Z_comp, Z_mean


def regression(comp, mean, stdev, inst, config):
    """decomposition.py:141-148."""
    M = np.dot(comp, comp.T)
    if not np.allclose(M, np.identity(M.shape[0]), atol=1e-5):
        det = np.linalg.det(M)
        print(f"WARNING: Computed basis is not orthonormal (determinant={det})")
    return linreg_lstsq(comp, mean, stdev, inst, config)


def compute(config, dump_name, instrumented_model):
    global B
    timestamp = lambda: datetime.datetime.now().strftime("%d.%m %H:%M")
    print(f"[{timestamp()}] Computing", dump_name.name)

    torch.manual_seed(0)
    np.random.seed(0)

    if not torch.cuda.is_available():
        raise RuntimeError("ganspace_amd.decomposition needs a HIP device (no CPU fallback)")
    device = torch.device("cuda", torch.cuda.current_device())
    layer_key = config.layer

    if instrumented_model is None:
        inst = get_instrumented_model(config.model, config.output_class, layer_key, device)
        model = inst.model
    else:
        print("Reusing InstrumentedModel instance")
        inst = instrumented_model
        model = inst.model
        inst.remove_edits()
        model.set_output_class(config.output_class)

    if config.use_w:
        print("Using W latent space")
        model.use_w()

    inst.retain_layer(layer_key)
    with torch.no_grad():
        model.partial_forward(model.sample_latent(1), layer_key)
    sample_shape = inst.retained_features()[layer_key].shape
    sample_dims = int(np.prod(sample_shape))
    print("Feature shape:", sample_shape)

    input_shape = inst.model.get_latent_shape()
    input_dims = int(inst.model.get_latent_dims())

    config.components = min(config.components, sample_dims)
    transformer = get_estimator(config.estimator, config.components, config.sparsity)
    if not transformer.batch_support:
        raise RuntimeError("only batch estimators run on the device path")

    X = None
    B = config.batch_size or get_max_batch_size(inst, device, layer_key)
    N = config.n // B * B
    print("B={}, N={}, dims={}, N/dims={:.1f}".format(B, N, sample_dims, N / sample_dims), flush=True)
    NB = max(B, max(2_000, 3 * config.components))

    torch.manual_seed(config.seed or SEED_SAMPLING)
    np.random.seed(config.seed or SEED_SAMPLING)

    # Same latents as the reference for a given (seed, B); kept in HBM, not host RAM
    n_lat = ((N + NB - 1) // B + 1) * B
    latents = torch.zeros((n_lat, *input_shape[1:]), dtype=torch.float32, device=device)
    with torch.no_grad():
        if hasattr(model, "z_spec") and hasattr(model, "latent_from_z"):
            # same stream consumption as n_lat // B calls of sample_latent(): one randint per mini-batch
            # (wrappers.py:168-169); the batches themselves are generated by parallel worker processes
            kind, zdim = model.z_spec
            seeds = [np.random.randint(np.iinfo(np.int32).max) for _ in range(n_lat // B)]
            zs = _zgen.generate(kind, seeds, B, zdim, getattr(model, "truncation", 1.0) if kind == "biggan" else 1.0)
            for i, z in enumerate(_progress(zs, desc="Sampling latents")):
                latents[i * B:(i + 1) * B] = model.latent_from_z(z)
        else:
            for i in _progress(range(n_lat // B), desc="Sampling latents"):
                latents[i * B:(i + 1) * B] = model.sample_latent(n_samples=B)

    samples_are_latents = layer_key in ["g_mapping", "style"] and inst.model.latent_space_name() == "W"

    canceled = False
    gi = 0
    try:
        X = torch.ones((NB, sample_dims), dtype=torch.float32, device=device)
        for gi in _progress(range(0, N, NB), desc=f"Fitting batches (NB={NB})"):
            for mb in range(0, NB, B):
                z = latents[gi + mb:gi + mb + B]
                if samples_are_latents:
                    batch = z.reshape((B, -1))
                else:
                    with torch.no_grad():
                        model.partial_forward(z, layer_key)
                    batch = inst.retained_features()[layer_key].reshape((B, -1))
                space_left = min(B, NB - mb)
                X[mb:mb + space_left] = batch[:space_left]
            if not transformer.fit_partial(X.reshape(-1, sample_dims)):
                break
    except KeyboardInterrupt:
        dump_name = dump_name.parent / dump_name.name.replace(f"n{N}", f"n{gi}")
        print(f'Saving current state to "{dump_name.name}" before exiting')
        canceled = True

    X_global_mean = transformer.transformer.mean_.reshape((1, sample_dims))
    # last block, centred with the final mean (decomposition.py:289-291); only its first 5000 rows
    # are ever used again (random-direction statistics, :312-316), so only those leave the device
    n_rand_samples = min(5000, X.shape[0])
    Xh = X[:n_rand_samples].cpu().numpy().astype(np.float32)
    Xh -= X_global_mean.astype(np.float32)

    X_comp, X_stdev, X_var_ratio = transformer.get_components()
    X_comp = np.array(X_comp, dtype=np.float32)
    assert X_comp.shape[1] == sample_dims and X_comp.shape[0] == config.components \
        and X_global_mean.shape[1] == sample_dims and X_stdev.shape[0] == config.components, "Invalid shape"

    if samples_are_latents:
        Z_comp = X_comp          # same array, as in the reference (:301-303): normalised in place below
        Z_global_mean = X_global_mean
    else:
        Z_comp, Z_global_mean = regression(X_comp, X_global_mean, X_stdev, inst, config)

    Z_comp /= np.linalg.norm(Z_comp, axis=-1, keepdims=True)

    random_dirs = get_random_dirs(config.components, int(np.prod(sample_shape)))
    X_stdev_random = np.dot(random_dirs, Xh.T).std(axis=1)

    X_comp = X_comp.reshape(-1, *sample_shape)
    X_global_mean = X_global_mean.reshape(sample_shape)
    Z_comp = Z_comp.reshape(-1, *input_shape)
    Z_global_mean = Z_global_mean.reshape(input_shape)

    lat_stdev = np.ones_like(X_stdev)
    if config.use_w:
        with torch.no_grad():
            samples = model.sample_latent(5000).reshape(5000, input_dims).detach().cpu().numpy()
        coords = np.dot(Z_comp.reshape(-1, input_dims), samples.T)
        lat_stdev = coords.std(axis=1)

    os.makedirs(dump_name.parent, exist_ok=True)
    np.savez_compressed(dump_name, **{
        "act_comp": X_comp.astype(np.float32),
        "act_mean": X_global_mean.astype(np.float32),
        "act_stdev": X_stdev.astype(np.float32),
        "lat_comp": Z_comp.astype(np.float32),
        "lat_mean": Z_global_mean.astype(np.float32),
        "lat_stdev": lat_stdev.astype(np.float32),
        "var_ratio": X_var_ratio.astype(np.float32),
        "random_stdevs": X_stdev_random.astype(np.float32),
    })

    if canceled:
        sys.exit(1)

    if instrumented_model is None:
        inst.close()
        del inst
        del model
    del X, latents
    torch.cuda.empty_cache()


def get_or_compute(config, model=None, submit_config=None, force_recompute=False):
    """decomposition.py:362-368."""
    if submit_config is None:
        wrkdir = os.environ.get("GANSPACE_RUN_DIR", str(Path(__file__).parent.parent.resolve()))
        submit_config = SimpleNamespace(run_dir_root=wrkdir, run_dir=wrkdir)
    return _compute(submit_config, config, model, force_recompute)


def _compute(submit_config, config, model=None, force_recompute=False):
    """decomposition.py:370-402: validation + cache-file naming."""
    basedir = Path(submit_config.run_dir)
    if config.n is None:
        raise RuntimeError("Must specify number of samples with -n=XXX")
    if model and not isinstance(model, InstrumentedModel):
        raise RuntimeError('Passed model has to be wrapped in "InstrumentedModel"')
    if config.use_w and "StyleGAN" not in config.model:
        raise RuntimeError(f"Cannot change latent space of non-StyleGAN model {config.model}")

    transformer = get_estimator(config.estimator, config.components, config.sparsity)
    dump_name = "{}-{}_{}_{}_n{}{}{}.npz".format(
        config.model.lower(),
        str(config.output_class).replace(" ", "_"),
        config.layer.lower(),
        transformer.get_param_str(),
        config.n,
        "_w" if config.use_w else "",
        f"_seed{config.seed}" if config.seed else "",
    )
    dump_path = basedir / "cache" / "components" / dump_name
    if not dump_path.is_file() or force_recompute:
        print("Not cached")
        t_start = datetime.datetime.now()
        compute(config, dump_path, model)
        print("Total time:", datetime.datetime.now() - t_start)
    return dump_path

"""Oracle restatement of the whole ``compute()`` pipeline on the CPU (TEST INFRASTRUCTURE).

Follows ``/root/reference/decomposition.py:150-358`` step by step with NumPy: seeding
(:157-158, :226-227), latent pre-sampling (:232-236), the block loop (:241-267) feeding the
SVD-recurrence oracle, finalisation (:288-329: mean, components, regression via
``scipy.linalg.lstsq(..., 'gelsd')`` :133, random-direction stdevs :312-316, ``lat_stdev``
:325-329) and returns the eight ``.npz`` arrays (:331-341).  ``features(z)`` maps a latent
mini-batch to the flattened activation (float64 math, cast to float32 like the device tensor).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

from . import zstream
from .ipca import IPCAEstimatorOracle

SEED_RANDOM_DIRS = 2
SEED_LINREG = 3


def random_dirs(components, dimensions):
    gen = np.random.RandomState(seed=SEED_RANDOM_DIRS)
    dirs = gen.normal(size=(components, dimensions))
    dirs /= np.sqrt(np.sum(dirs ** 2, axis=1, keepdims=True))
    return dirs.astype(np.float32)


def run(n, batch_size, components, features, latent_to_primary=None, use_w=False, seed=None,
        latent_kind="stylegan", kind="svd"):
    """``latent_to_primary``: z -> primary latent (the W mapping when ``use_w``; identity otherwise).
    ``features``: primary latent -> activation ``[B, d]`` (None when samples are the latents)."""
    B, N, NB, n_lat, n_blocks = zstream.loop_plan(n, batch_size, components)
    n_batches = n_lat // B
    # one extra draw from the global stream for the lat_stdev sample (decomposition.py:327)
    seeds = zstream.batch_seeds(n_batches + 1, seed)
    zfn = zstream.stylegan_z_batch if latent_kind == "stylegan" else zstream.biggan_z_batch
    prim = (lambda z: z) if latent_to_primary is None else latent_to_primary
    latents = np.concatenate([np.asarray(prim(zfn(s, B)), dtype=np.float32) for s in seeds[:n_batches]], axis=0)

    est = IPCAEstimatorOracle(components, kind)
    X = None
    for X in zstream.iter_blocks(latents, n, batch_size, components, features):
        if not est.fit_partial(X):
            break
    t = est.transformer
    mean = np.asarray(t.mean_, dtype=np.float64).reshape(1, -1)
    Xc = X.astype(np.float32) - mean.astype(np.float32)
    comp, stdev, ratio = est.get_components()
    comp = np.asarray(comp, dtype=np.float64)
    d = comp.shape[1]

    if features is None:
        Z_comp, Z_mean = comp.copy(), mean
    else:
        # linreg_lstsq (:77-139): fresh latents from the global stream re-seeded to 3
        # (the first draw after re-seeding is consumed by get_latent_dims() -> sample_latent(1), :87)
        rs_seeds = zstream.batch_seeds(1 + max(10_000, n) // B, SEED_LINREG)[1:]
        A_rows, Z_rows = [], []
        for s in rs_seeds:
            z = np.asarray(prim(zfn(s, B)), dtype=np.float32)
            act = np.asarray(features(z), dtype=np.float64).reshape(B, -1)
            A_rows.append(((act - mean) @ comp.T) / stdev)
            Z_rows.append(z.reshape(B, -1).astype(np.float64))
        A, Z = np.concatenate(A_rows), np.concatenate(Z_rows)
        M_t = scipy.linalg.lstsq(A, Z, lapack_driver="gelsd")[0]
        Z_comp, Z_mean = M_t[:components], Z.mean(axis=0, keepdims=True)
    Z_comp = Z_comp / np.linalg.norm(Z_comp, axis=-1, keepdims=True)

    rd = random_dirs(components, d)
    n_rand = min(5000, Xc.shape[0])
    random_stdevs = np.dot(rd, Xc[:n_rand].T).std(axis=1)

    lat_stdev = np.ones_like(stdev)
    if use_w:
        samples = np.asarray(prim(zfn(seeds[n_batches], 5000)), dtype=np.float32).reshape(5000, -1)
        lat_stdev = np.dot(Z_comp, samples.T.astype(np.float64)).std(axis=1)

    return dict(act_comp=comp.astype(np.float32), act_mean=mean.astype(np.float32),
                act_stdev=np.asarray(stdev, np.float32), lat_comp=Z_comp.astype(np.float32),
                lat_mean=np.asarray(Z_mean, np.float32), lat_stdev=np.asarray(lat_stdev, np.float32),
                var_ratio=np.asarray(ratio, np.float32), random_stdevs=random_stdevs.astype(np.float32))

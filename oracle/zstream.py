"""Oracle restatement of the latent stream and block indexing (TEST INFRASTRUCTURE).

Follows ``/root/reference/decomposition.py:198-267`` (batch/block arithmetic and
the "Sampling latents" / "Fitting batches" loops) and the per-batch seeding of
``StyleGAN2.sample_latent`` (``/root/reference/models/wrappers.py:167-179``) and
``BigGAN.sample_latent`` (``wrappers.py:562-569`` ->
``models/biggan/pytorch_biggan/pytorch_pretrained_biggan/utils.py:21-33``).
"""
from __future__ import annotations

import numpy as np

SEED_SAMPLING = 1        # decomposition.py:34


def loop_plan(n: int, batch_size: int, components: int):
    """(B, N, NB, n_lat, n_blocks) exactly as ``compute()`` derives them.

    decomposition.py:198 (B), :201 (N = n // B * B), :220 (NB), :232 (n_lat),
    :245 (ceil(N / NB) blocks).
    """
    B = int(batch_size)
    N = n // B * B
    NB = max(B, max(2000, 3 * components))
    n_lat = ((N + NB - 1) // B + 1) * B
    n_blocks = (N + NB - 1) // NB
    return B, N, NB, n_lat, n_blocks


def batch_seeds(n_batches: int, seed=None):
    """The per-mini-batch seeds drawn from the *global legacy* NumPy stream.

    ``np.random.seed(config.seed or 1)`` (decomposition.py:227) followed by one
    ``np.random.randint(np.iinfo(np.int32).max)`` per ``sample_latent`` call
    (wrappers.py:168-169).  A private ``RandomState`` with the same seed yields
    the same stream without touching global state.
    """
    rs = np.random.RandomState(seed or SEED_SAMPLING)
    hi = np.iinfo(np.int32).max
    return [int(rs.randint(hi)) for _ in range(n_batches)]


def stylegan_z_batch(seed: int, n_samples: int, dim: int = 512) -> np.ndarray:
    """z of one StyleGAN2 mini-batch, float32 [n, dim] (wrappers.py:171-174)."""
    rng = np.random.RandomState(seed)
    return rng.standard_normal(dim * n_samples).reshape(n_samples, dim).astype(np.float32)


def biggan_z_batch(seed: int, n_samples: int, dim: int = 128, truncation: float = 1.0) -> np.ndarray:
    """Truncated-normal z of one BigGAN mini-batch (biggan/.../utils.py:31-33)."""
    from scipy.stats import truncnorm
    state = np.random.RandomState(seed)
    v = truncnorm.rvs(-2, 2, size=(n_samples, dim), random_state=state).astype(np.float32)
    return truncation * v


def sample_all_latents(n: int, batch_size: int, components: int, seed=None,
                       kind: str = "stylegan", dim=None):
    """The host ``latents`` array of decomposition.py:232-236 (Z space)."""
    B, N, NB, n_lat, _ = loop_plan(n, batch_size, components)
    seeds = batch_seeds(n_lat // B, seed)
    if kind == "stylegan":
        dim = dim or 512
        parts = [stylegan_z_batch(s, B, dim) for s in seeds]
    elif kind == "biggan":
        dim = dim or 128
        parts = [biggan_z_batch(s, B, dim) for s in seeds]
    else:
        raise ValueError(kind)
    return np.concatenate(parts, axis=0)


def block_rows(gi: int, NB: int, B: int):
    """Row indices of ``latents`` that end up in IPCA block starting at ``gi``.

    decomposition.py:246-261: mini-batch ``mb`` evaluates rows
    ``gi+mb : gi+mb+B`` and keeps the first ``min(B, NB-mb)`` of them, so the
    block is simply ``latents[gi : gi+NB]`` (the tail mini-batch is evaluated on
    B rows but truncated).
    """
    return np.arange(gi, gi + NB)


def iter_blocks(latents: np.ndarray, n: int, batch_size: int, components: int, feature_fn=None):
    """Yield the ``X[NB, d]`` float32 blocks ``fit_partial`` receives.

    ``feature_fn`` maps a latent mini-batch ``[B, ...]`` to features ``[B, d]``
    (identity when the samples are the latents, decomposition.py:249-251).
    """
    B, N, NB, _, _ = loop_plan(n, batch_size, components)
    for gi in range(0, N, NB):
        rows = []
        for mb in range(0, NB, B):
            z = latents[gi + mb: gi + mb + B]
            f = z.reshape(B, -1) if feature_fn is None else np.asarray(feature_fn(z)).reshape(B, -1)
            rows.append(f[: min(B, NB - mb)])
        yield np.concatenate(rows, axis=0).astype(np.float32, copy=False)

"""Deterministic inputs for the golden fixtures (shared by make_golden.py and tests).

The fixtures store only the reference's *outputs*; the inputs are rebuilt here
from NumPy legacy ``RandomState`` streams (bit-stable across NumPy versions) so
that the HIP path, the oracle and the reference all see identical float32 rows.
"""
import numpy as np

# name -> parameters.  ``blocks`` are the row counts handed to fit_partial in order;
# ``ncheck`` = number of leading components that are well conditioned (compared by signed cosine).
IPCA_CASES = {
    # tiny, d not a multiple of any tile size
    "d64_k8": dict(seed=11, d=64, k=8, rank=24, decay=1.25, ncheck=8, blocks=[300, 300, 300, 300], mean_scale=0.3, noise=0.05),
    # BASELINE config 1 shape: d=512, k=20, NB=2000, 5 blocks (N=9728 -> 10 000 rows)
    "d512_k20": dict(seed=12, d=512, k=20, rank=96, decay=1.25, ncheck=20, blocks=[2000] * 5, mean_scale=0.2, noise=0.05),
    # BASELINE config 2 shape (3 of the 100 blocks): d=512, k=80, NB=10 000
    "d512_k80_nb10000": dict(seed=13, d=512, k=80, rank=160, decay=1.04, ncheck=80, blocks=[10000] * 3, mean_scale=0.2, noise=0.05),
    # ragged block sizes, k == d (IPCA is exact PCA then), d not a multiple of 32
    "d200_k200_ragged": dict(seed=14, d=200, k=200, rank=200, decay=1.06, ncheck=100, blocks=[700, 333, 1024], mean_scale=0.5, noise=0.1),
    # |mean| >> stdev: the catastrophic-cancellation trap of raw moments (BigGAN gen_z-like bias)
    "d96_k12_bigmean": dict(seed=15, d=96, k=12, rank=32, decay=1.25, ncheck=12, blocks=[500, 500, 500], mean_scale=40.0, noise=0.05),
    # feat_dim >> block rows: the small-side path (cfg3 / cfg5 shape in miniature)
    "d3000_k12_highd": dict(seed=17, d=3000, k=12, rank=40, decay=1.12, ncheck=12, blocks=[300, 300, 300],
                            mean_scale=1.0, noise=0.05),
    # affine in a 16-d latent: covariance rank 16 < k (BigGAN gen_z is affine in z)
    "d128_k24_lowrank": dict(seed=16, d=128, k=24, rank=16, decay=1.15, ncheck=16, blocks=[400, 400, 400], mean_scale=1.0, noise=0.0),
}


def _mixing(case):
    rs = np.random.RandomState(case["seed"])
    d, r = case["d"], case["rank"]
    A = rs.standard_normal((r, d))
    # decaying, well separated spectrum: scale_i = decay^-i; ``ncheck`` leading components sit
    # well above the isotropic noise floor and are compared vector-by-vector
    A *= (case["decay"] ** -np.arange(r))[:, None] * 3.0
    mu = rs.standard_normal(d) * case["mean_scale"]
    return rs, A, mu


def ipca_blocks(case):
    """Yield float32 ``[m, d]`` blocks, one per ``fit_partial`` call."""
    rs, A, mu = _mixing(case)
    for m in case["blocks"]:
        Z = rs.standard_normal((m, case["rank"]))
        X = Z @ A + mu
        if case["noise"] > 0:
            X = X + case["noise"] * rs.standard_normal((m, case["d"]))
        yield X.astype(np.float32)


MAPPING_CASE = dict(seed=21, rows=64, dim=512, layers=8, lr_mul=0.01)


def mapping_weights(case=MAPPING_CASE):
    """EqualLinear parameters: ``weight = randn(out, in) / lr_mul``; small non-zero biases."""
    rs = np.random.RandomState(case["seed"])
    L, d = case["layers"], case["dim"]
    W = (rs.standard_normal((L, d, d)) / case["lr_mul"]).astype(np.float32)
    # the reference initialises biases to zero; use zeros so G_mapping (which applies the
    # bias inside the un-gained lrelu) and EqualLinear agree exactly in form
    b = np.zeros((L, d), dtype=np.float32)
    return W, b


def mapping_z(case=MAPPING_CASE):
    rs = np.random.RandomState(case["seed"] + 1000)
    return rs.standard_normal((case["rows"], case["dim"])).astype(np.float32)

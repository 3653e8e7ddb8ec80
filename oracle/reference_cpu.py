"""The reference's CPU arithmetic, instantiated exactly as the reference does
(TEST / BASELINE INFRASTRUCTURE - never imported by ganspace_amd/).

``/root/reference/estimators.py:59`` builds
``IncrementalPCA(n_components, whiten=False, batch_size=max(100, 2*n_components))`` and
``fit_partial`` (:68-76) calls ``partial_fit`` then casts ``n_samples_seen_`` to int64.
scikit-learn is the third-party dependency that holds the arithmetic; it is
installed in the image (1.7.2), so ``bench.py``'s ``cpu_baseline`` leg times *it* - the very
code the reference executes - rather than the NumPy restatement in ``oracle/ipca.py``.
"""
import time

import numpy as np


def make_reference_ipca(n_components: int):
    from sklearn.decomposition import IncrementalPCA
    return IncrementalPCA(n_components, whiten=False, batch_size=max(100, 2 * n_components))


def time_reference_fit(blocks, n_components: int):
    """Run the reference "Fitting batches" arithmetic on host float32 blocks.

    Returns ``(ipca, seconds, samples)``; ``seconds`` covers the partial_fit calls only
    (decomposition.py:263-264), the blocks being already in host memory.
    """
    ipca = make_reference_ipca(n_components)
    t = 0.0
    n = 0
    for X in blocks:
        t0 = time.perf_counter()
        ipca.partial_fit(X)
        ipca.n_samples_seen_ = np.int64(ipca.n_samples_seen_)
        t += time.perf_counter() - t0
        n += X.shape[0]
    return ipca, t, n


def host_threads():
    import os
    try:
        from threadpoolctl import threadpool_info
        blas = [i.get("num_threads", 0) for i in threadpool_info() if i.get("user_api") == "blas"]
        if blas:
            return max(blas)
    except Exception:
        pass
    return os.cpu_count() or 1

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


def time_reference_fit(blocks, n_components: int, per_block: bool = False):
    """Run the reference "Fitting batches" arithmetic on host float32 blocks.

    Returns ``(ipca, seconds, samples)`` - or ``(ipca, [seconds per block])`` with ``per_block`` -; the time covers
    the partial_fit calls only (decomposition.py:263-264), the blocks being already in host memory.
    """
    ipca = make_reference_ipca(n_components)
    times = []
    n = 0
    for X in blocks:
        t0 = time.perf_counter()
        ipca.partial_fit(X)
        ipca.n_samples_seen_ = np.int64(ipca.n_samples_seen_)
        times.append(time.perf_counter() - t0)
        n += X.shape[0]
    if per_block:
        return ipca, times
    return ipca, float(sum(times)), n


def blas_threads(n):
    """Context manager limiting the BLAS / LAPACK pools to ``n`` threads (no-op without threadpoolctl)."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=int(n), user_api="blas")
    except Exception:
        import contextlib
        return contextlib.nullcontext()


def pick_blas_threads(blocks, n_components: int, candidates=(8, 16, 32, 64)):
    """Steady-state seconds per block of the reference arithmetic for a few BLAS thread counts (each: one warm-up
    block, then the remaining ``blocks``): returns ``(fastest count, {count: seconds per block})``."""
    import os
    top = max(1, min(host_threads(), os.cpu_count() or 1))
    cands = sorted({c for c in candidates if c <= top} | {top})
    tried = {}
    for c in cands:
        with blas_threads(c):
            _, times = time_reference_fit(blocks, n_components, per_block=True)
        tried[str(c)] = round(float(np.mean(times[1:])) if len(times) > 1 else times[0], 4)
    best = min(tried, key=tried.get)
    return int(best), tried


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

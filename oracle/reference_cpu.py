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


def _wide_baseline_main(argv):
    """``python -m oracle.reference_cpu <feat_dim> <k> <blocks> <blas_threads> <out.json> [out_components.npy]``

    bench.py's wide-feature CPU baseline as a child process (it runs beside the GPU legs of the bench: ``blocks`` x
    26 s at d = 131 072 would otherwise be most of the run): scikit-learn's ``IncrementalPCA.partial_fit`` - the
    reference's arithmetic, ``/root/reference/estimators.py:68-76`` - on the CPU-seeded synthetic blocks of SURVEY.md 8d
    item 5 (``oracle.smallside_torch.lowrank_plus_noise_blocks(device="cpu")``: the parent regenerates the same blocks
    for the cosine check), seconds per block to ``out.json``."""
    import json
    import os
    import torch
    from oracle.smallside_torch import lowrank_plus_noise_blocks
    d, k, nb, threads, out = int(argv[0]), int(argv[1]), int(argv[2]), int(argv[3]), argv[4]
    torch.set_num_threads(max(1, min(threads, 16)))
    sk = make_reference_ipca(k)
    per = []
    with blas_threads(threads):
        for X in lowrank_plus_noise_blocks(d, nb, rows=2000, device="cpu"):
            hb = X.numpy()
            t0 = time.perf_counter()
            sk.partial_fit(hb)
            per.append(time.perf_counter() - t0)
            with open(out + ".partial", "w") as f:           # (a parent that runs out of patience reads what is there)
                json.dump({"seconds_per_block": per, "blas_threads": threads, "feat_dim": d, "done": False}, f)
    if len(argv) > 5:
        np.save(argv[5], sk.components_.astype(np.float32))
    with open(out, "w") as f:
        json.dump({"seconds_per_block": per, "blas_threads": threads, "feat_dim": d, "done": True,
                   "host_cpu_count": os.cpu_count()}, f)


if __name__ == "__main__":
    import sys
    _wide_baseline_main(sys.argv[1:])

"""Kernel-level view of the eigensolve (cfg2: d = 512, k = 80): run under
``rocprofv3 --kernel-trace --stats``; prints host-side times of `reps` exact-mode finalizes (cold top-k solve)
and of a faithful-mode run (one warm-started solve per block).

    python tools/finalize_trace.py [blocks] [reps] [mode]      mode: exact | faithful | both
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ganspace_amd.estimators import IPCAEstimator
from ganspace_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "both"
blocks = bench.make_blocks(-(-nb // 5) * 5, dev)[0]
blocks = blocks[:nb]
if mode in ("exact", "both"):
    for rep in range(reps):
        est = IPCAEstimator(80, "exact")
        for b in blocks:
            est.fit_partial(b)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        est.get_components()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        h = est.transformer._h
        print(f"exact finalize {1e3*(t2-t1):.3f} ms, products={lib.gs_ipca_last_mults(h)} sweeps={lib.gs_ipca_last_sweeps(h)}", flush=True)
if mode in ("faithful", "both"):
    for rep in range(max(1, reps // 2)):
        est = IPCAEstimator(80, "faithful")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        per = []
        for b in blocks:
            t0 = time.perf_counter()
            est.fit_partial(b)
            torch.cuda.synchronize()
            per.append(1e3 * (time.perf_counter() - t0))
        t2 = time.perf_counter()
        h = est.transformer._h
        print(f"faithful {nb} blocks {1e3*(t2-t1):.2f} ms; per block ms: " + " ".join(f"{p:.2f}" for p in per) +
              f"; products(last)={lib.gs_ipca_last_mults(h)} sweeps(last)={lib.gs_ipca_last_sweeps(h)}", flush=True)

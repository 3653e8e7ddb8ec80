"""Where a short job's fit loop goes besides kernels: gs_ipca_create (allocations), the first update, the finalize.
    python tools/create_probe.py"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganspace_amd.estimators import IPCAEstimator
dev = torch.device("cuda", 0)
X = torch.randn(1_000_000, 512, device=dev)
def sync(): torch.cuda.synchronize()
for mode in ("exact", "exact", "exact", "faithful", "faithful"):
    sync(); t0 = time.perf_counter()
    est = IPCAEstimator(80, mode=mode, device=dev)
    est.transformer._ensure(512)
    sync(); t1 = time.perf_counter()
    if mode == "exact":
        for lo in range(0, 1_000_000, 80_000):
            est.fit_partial(X[lo:lo + 80_000], resident=True)
    else:
        for lo in range(0, 1_000_000, 10_000):
            est.fit_partial(X[lo:lo + 10_000])
    sync(); t2 = time.perf_counter()
    est.get_components()
    sync(); t3 = time.perf_counter()
    est.transformer.close()
    sync(); t4 = time.perf_counter()
    print(f"{mode:9s} create {1e3*(t1-t0):7.2f} ms   updates {1e3*(t2-t1):7.2f} ms   results {1e3*(t3-t2):7.2f} ms   destroy {1e3*(t4-t3):7.2f} ms",
          flush=True)

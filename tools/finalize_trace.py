"""Kernel-level view of ONE exact-mode finalize (cfg2: d = 512, k = 80): run under
rocprofv3 --kernel-trace --stats; the update launches are done before the marker kernel count is taken."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ganspace_amd.estimators import IPCAEstimator
from ganspace_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lat = bench.make_latents(nb, dev, 0)
est = IPCAEstimator(80, "exact")
for i in range(nb):
    est.fit_partial(lat[i * 10000:(i + 1) * 10000])
torch.cuda.synchronize(); t1 = time.perf_counter()
est.get_components()
torch.cuda.synchronize(); t2 = time.perf_counter()
h = est.transformer._h
print(f"finalize {1e3*(t2-t1):.2f} ms, mults={lib.gs_ipca_last_mults(h)} sweeps={lib.gs_ipca_last_sweeps(h)}", flush=True)

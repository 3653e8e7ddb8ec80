import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ganspace_amd.estimators import IPCAEstimator
from ganspace_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
lat = bench.make_latents(10, dev, 0)
for mode in ("exact", "faithful"):
    for rep in range(2):
        est = IPCAEstimator(80, mode)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10):
            est.fit_partial(lat[i * 10000:(i + 1) * 10000])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        est.get_components()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        h = est.transformer._h
        print(f"{mode}: 10 updates {1e3*(t1-t0):.2f} ms, finalize {1e3*(t2-t1):.2f} ms, mults={lib.gs_ipca_last_mults(h)} sweeps={lib.gs_ipca_last_sweeps(h)}", flush=True)

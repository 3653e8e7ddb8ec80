"""The faithful (`--est=ipca`) cfg2 job of bench.py's `faithful_mode_same_job` alone: 100 blocks of 10 000 W-space rows, no
per-block synchronisation, best of five runs.  python tools/faithful_job_probe.py [blocks]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ganspace_amd.estimators import IPCAEstimator
dev = torch.device("cuda", 0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
blocks = bench.make_blocks(-(-nb // 5) * 5, dev)[0][:nb]
ef = IPCAEstimator(80, "faithful")
for b in blocks[:20]:
    ef.fit_partial(b)
ef.get_components()
runs = []
for rep in range(5):
    ef = IPCAEstimator(80, "faithful")
    ef.transformer._ensure(512)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in blocks:
        ef.fit_partial(b)
    ef.get_components()
    torch.cuda.synchronize(); runs.append(time.perf_counter() - t0)
print(f"faithful {nb} blocks: best {min(runs)*1e3:.2f} ms = {min(runs)/nb*1e3:.4f} ms per block; runs {[round(r*1e3,2) for r in runs]}")

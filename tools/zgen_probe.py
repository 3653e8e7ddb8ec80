"""Where the latent pre-sampling time goes (run on the GPU box): pinned-ring allocation, native generation alone for a
few thread counts, and the whole _presample of the cfg2 job."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ganspace_amd import _zgen, _lib
_lib.load()
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
for mb in (360, 720, 1440):
    t0 = time.perf_counter(); x = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, pin_memory=True); dt = time.perf_counter() - t0
    print(f"pin {mb} MB: {dt*1e3:.0f} ms", flush=True); del x
seeds = [int(s) for s in np.random.RandomState(1).randint(0, 2**31 - 1, size=101)]
for thr in (16, 32, 64, 101):
    _zgen._RING_CACHE.clear()
    t0 = time.perf_counter()
    st = _zgen.NativeNormalStream(seeds, 10000, 512, threads=thr, pinned=False)
    t1 = time.perf_counter()
    for i, z in st:
        st.release(i + 1)
    st.close()
    print(f"native gen only, {thr} threads: start {1e3*(t1-t0):.0f} ms, total {1e3*(time.perf_counter()-t0):.0f} ms", flush=True)
import bench
for rep in range(2):
    t0 = time.perf_counter(); blocks, steps, t, model = bench.make_blocks(100, dev); print(f"make_blocks rep {rep}: {time.perf_counter()-t0:.2f} s", flush=True)

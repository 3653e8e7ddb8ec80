"""Device z generator: kernel time for the cfg2 / cfg3 seed lists, and the pre-sampling phase (bench.make_blocks) with the
device generator against the host thread pool (GANSPACE_ZGEN=host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ganspace_amd import _zgen, _lib
_lib.load()
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
for kind, nseeds, n, dim in (("stylegan", 101, 10000, 512), ("stylegan", 256, 10000, 512), ("biggan", 501, 2000, 128)):
    seeds = [int(s) for s in np.random.RandomState(1).randint(0, 2**31 - 1, size=nseeds)]
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        last = None
        for i, z in _zgen.device_batches(kind, seeds, n, dim, dev):
            last = z
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"device zgen {kind} {nseeds} seeds x {n} x {dim}: {dt*1e3:.1f} ms ({nseeds*n*dim/dt/1e9:.2f} G values/s)", flush=True)
import bench
for mode in ("device", "host", "device"):
    os.environ["GANSPACE_ZGEN"] = mode
    t0 = time.perf_counter(); blocks, steps, t, model = bench.make_blocks(100, dev)
    print(f"make_blocks [{mode}]: T_sample {t:.3f} s (wall {time.perf_counter()-t0:.2f} s incl. model)", flush=True)
    del blocks, steps, model
    torch.cuda.empty_cache()

"""Is the duration of the Gram kernels a function of the DATA?  Same process, same launches: the bench's W-space rows
(mapping-network output), a fresh copy of them, N(0,1) rows, N(mean_c, std_c) rows with the W columns' moments."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ganspace_amd import _lib
from ganspace_amd.estimators import IPCAEstimator
lib = _lib.load()
dev = torch.device("cuda", 0)
bench.make_blocks(100, dev)
W = bench.RESIDENT_ROWS
mu, sd = W.mean(0), W.std(0)
sets = {"W rows (bench)": W, "W rows, fresh copy": W.clone(), "randn": torch.randn_like(W),
        "randn * std_c + mean_c": torch.randn_like(W) * sd + mu, "zeros": torch.zeros_like(W)}
for prec, rows in (("f32", 131072), ("bf16", 1000000), ("bf16x3", 131072)):
    est = IPCAEstimator(80, "exact", precision=prec); est.transformer._ensure(512)
    for rep in range(2):
        for name, X in sets.items():
            us, rt = bench.gram_kernel_us(lib, _lib, est, X[:rows])
            print(f"{prec:7s} rows={rt:8d} {name:26s} {us:8.1f} us  {rt*2048/us/1e3:7.0f} GB/s", flush=True)

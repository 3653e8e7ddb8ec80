"""One process that exercises the Gram kernel in its three contraction modes and the mapping network, for a
rocprofv3 --kernel-trace --stats run (profiles/r01_modes_kernel_stats.csv)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ganspace_amd import ops
from ganspace_amd.estimators import IPCAEstimator
dev = torch.device("cuda", 0)
X = torch.randn(10000, 512, device=dev)
for prec in ("f32", "bf16x6", "bf16x3"):
    est = IPCAEstimator(80, "exact", precision=prec)
    for i in range(100):
        est.fit_partial(X)
    est.get_components()
W, b = bench.make_mapping_weights(dev)
z = torch.randn(10000, 512, device=dev)
for i in range(20):
    ops.mapping_forward(z, W, b)
torch.cuda.synchronize()

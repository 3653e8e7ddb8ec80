import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganspace_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
B = torch.randn(2000, n, generator=g, dtype=torch.float64) * (1.02 ** -torch.arange(n, dtype=torch.float64))
A = (B.T @ B).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w, V, sweeps = ops.eigh_sym(A)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n} eigh {dt*1e3:.2f} ms sweeps={sweeps}")
wr = torch.linalg.eigvalsh(A.cpu()).flip(0)
print("max rel eig err", ((w.cpu() - wr).abs().max() / wr[0]).item())

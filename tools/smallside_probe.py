import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganspace_amd.estimators import get_estimator
from ganspace_amd import _lib
d = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 80
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 3
prec = sys.argv[5] if len(sys.argv) > 5 else "f32"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(128, d, device=dev, generator=g) * (1.03 ** -torch.arange(128, device=dev))[:, None]
from ganspace_amd.estimators import IPCAEstimator
est = IPCAEstimator(k, "faithful", precision=prec)
lib = _lib.load()
for i in range(nb):
    X = torch.randn(m, 128, device=dev, generator=g) @ A + 0.05 * torch.randn(m, d, device=dev, generator=g) + 0.3
    torch.cuda.synchronize(); t0 = time.perf_counter()
    est.fit_partial(X)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{prec} d={d} m={m} k={k} block {i}: {dt*1e3:.1f} ms  ({m/dt:.0f} samples/s)  sweeps={lib.gs_ipca_last_sweeps(est.transformer._h)} mults={lib.gs_ipca_last_mults(est.transformer._h)}", flush=True)

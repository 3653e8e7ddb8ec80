import sys, time, torch
sys.path.insert(0, '/root/repo')
from ganspace_amd.estimators import IPCAEstimator
from ganspace_amd import _lib
lib = _lib.load()
dev = torch.device('cuda', 0)
X = torch.randn(10000, 512, device=dev)
for rep in range(2):
    est = IPCAEstimator(80, 'exact')
    for i in range(10): est.fit_partial(X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    comp, sd, vr = est.get_components()
    torch.cuda.synchronize()
    h = est.transformer._h
    print('white noise finalize %.1f ms mults=%d sweeps=%d' % ((time.perf_counter() - t0) * 1e3, lib.gs_ipca_last_mults(h), lib.gs_ipca_last_sweeps(h)))
import numpy as np
G = (X.double().T @ X.double()).cpu().numpy() * 10
Xc = X.double() - X.double().mean(0)
C = (Xc.T @ Xc).cpu().numpy() * 10
w, V = np.linalg.eigh(C)
V = V[:, ::-1][:, :80].T
# residual check: comp rows are eigenvectors of C
R = comp.astype(np.float64) @ C - (np.sum((comp.astype(np.float64) @ C) * comp, axis=1))[:, None] * comp
print('max residual / lambda1 = %.2e' % (np.abs(R).max() / w[-1]))

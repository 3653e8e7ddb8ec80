"""Faithful and exact estimators on 30 000-row blocks of d = 512 (wide launches inside gram_update): components with the
second-stream fold against the run with GS_GRAM_NO_AUX_FOLD=1 (argument: output .npy)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from ganspace_amd.estimators import IPCAEstimator
g = torch.Generator(device="cuda").manual_seed(7)
A = torch.randn(96, 512, device="cuda", generator=g) * (1.07 ** -torch.arange(96, device="cuda"))[:, None]
out = []
for mode in ("faithful", "exact"):
    est = IPCAEstimator(40, mode)
    for b in range(5):
        X = torch.randn(30000, 96, device="cuda", generator=g) @ A + 0.05 * torch.randn(30000, 512, device="cuda", generator=g) + 0.3
        assert est.fit_partial(X)
    c, s, r = est.get_components()
    out.append(np.concatenate([c.ravel(), s, r, est.transformer.mean_]))
np.save(sys.argv[1], np.concatenate(out))
print("saved", sys.argv[1])

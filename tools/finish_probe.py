import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ganspace_amd import _zgen, ops
dev = torch.device("cuda", 0)
head = torch.randn(5000, 512, device=dev)
def T(name, f, reps=3):
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = f(); torch.cuda.synchronize()
        print(f"{name}: {1e3*(time.perf_counter()-t0):.2f} ms", flush=True)
    return out
def gen():
    (_, d), = _zgen.device_batches("stylegan", [17], 80, 512, dev)
    return d
dirs = T("device_batches 80x512", gen)
dn = T("normalise", lambda: dirs / torch.linalg.norm(dirs, dim=1, keepdim=True))
pr = T("project_rows", lambda: ops.project_rows(head, dn))
T("double.std.cpu", lambda: pr.double().std(dim=0, unbiased=False).cpu().numpy())
def gen2():
    (_, d), = _zgen.device_batches("stylegan", [18], 5000, 512, dev)
    return d
T("device_batches 5000x512", gen2)
import tempfile
rec = {k: np.random.randn(80, 512).astype(np.float32) for k in ("a", "b")}
def wr():
    with tempfile.TemporaryDirectory() as td:
        np.savez_compressed(os.path.join(td, "x.npz"), **rec)
T("savez_compressed 2x80x512", wr)

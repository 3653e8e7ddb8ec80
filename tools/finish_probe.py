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
from ganspace_amd import _npz
big = {"act_comp": np.random.randn(80, 32768).astype(np.float32), "lat_comp": np.random.randn(80, 128).astype(np.float32)}
def wr_np():
    with tempfile.TemporaryDirectory() as td:
        np.savez_compressed(os.path.join(td, "x.npz"), **big)
def wr_par():
    with tempfile.TemporaryDirectory() as td:
        _npz.savez_compressed(os.path.join(td, "x.npz"), **big)
T("np.savez_compressed 80x32768", wr_np)
T("_npz.savez_compressed 80x32768 (deflate pieces on a thread pool)", wr_par)
for th in (2, 4, 8, 16, 32):
    def wr_t():
        with tempfile.TemporaryDirectory() as td:
            _npz.savez_compressed(os.path.join(td, "x.npz"), threads=th, **big)
    T(f"   ... {th} threads", wr_t, reps=2)

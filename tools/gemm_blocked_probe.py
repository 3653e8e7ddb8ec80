"""gs_gemm_blocked_nt against gs_linear_forward on the product shapes of the path that still use the register-staged kernel:
a mapping layer (10 000 x 512 x 512), BigGAN gen_z (2000 x 32 768 x 256) and the conv GEMM of convs.2 (128 000 x 512 x 4608)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ganspace_amd import ops
dev = torch.device("cuda", 0)
def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, K in (("mapping layer", 10000, 512, 512), ("gen_z", 2000, 32768, 256), ("convs.2 GEMM", 128000, 512, 4608)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
    t_lin = timed(lambda: ops.linear_forward(x, w, None))
    xb, wb = ops.block_rows(x), ops.block_rows(w)
    t_blk = timed(lambda: ops.gemm_blocked_nt(xb, M, wb, N, K))
    t_pre = timed(lambda: ops.block_rows(x))
    fl = 2.0 * M * N * K
    print(f"{name}: linear_forward {t_lin:.1f} us ({fl/t_lin/1e6:.1f} TF) | gemm_blocked_nt {t_blk:.1f} us ({fl/t_blk/1e6:.1f} TF) + block_rows(x) {t_pre:.1f} us", flush=True)

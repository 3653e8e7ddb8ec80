"""Time the mapping network (8 x [rows x 512 x 512] f32-MFMA layers, one `linear_act_fast_kernel` launch per layer) per call
length, and the gen_z-shaped Linear, on the device.
    python tools/mapping_probe.py [rows ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganspace_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
W = torch.randn(8, 512, 512, device=dev) / 0.01
b = torch.zeros(8, 512, device=dev)
rows_list = [int(a) for a in sys.argv[1:]] or [10000, 16384, 24576, 32768, 40000, 65536, 80000, 131072, 520000]
# with the measurement build (GANSPACE_HIP_LIB=.../lib_measure/...) the launch-time knobs select the kernel: A/B in one process
variants = [("default", {})]
if "lib_measure" in os.environ.get("GANSPACE_HIP_LIB", ""):
    variants += [("per-tile workgroups", {"GS_LINEAR_PERSIST": "0"}), ("one pipeline per CU", {"GS_LINEAR_PERSIST": "1"})]
for rows in rows_list:
    z = torch.randn(rows, 512, device=dev)
    out = torch.empty_like(z)
    iters = max(3, min(20, int(2e6 // rows)))
    outs = []
    for name, env in variants:
        os.environ.update(env)
        best = None
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(iters): ops.mapping_forward(z, W, b, out=out)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
            best = dt if best is None else min(best, dt)
        for k in env: del os.environ[k]
        outs.append(out.clone())
        fl = 8 * 2 * rows * 512 * 512
        print(f"mapping {rows:7d} x 512, 8 layers [{name}]: {best*1e6:9.0f} us  ({fl/best/1e12:6.1f} TF/s = "
              f"{fl/best/157.3e12:.3f} of peak, {best/8*1e6*10000/rows:6.1f} us per 10 000-row layer equivalent)", flush=True)
    if len(outs) > 1:
        print("   bit-identical across variants:", all(torch.equal(outs[0], o) for o in outs[1:]), flush=True)
    del z, out, outs
x = torch.randn(2000, 256, device=dev); Wg = torch.randn(32768, 256, device=dev) * 0.05; bg = torch.zeros(32768, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): y = ops.linear_forward(x, Wg, bg)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"gen_z 2000x256 -> 32768: {dt*1e6:.0f} us  ({2*2000*256*32768/dt/1e12:.1f} TF/s)")
ref = torch.nn.functional.linear(x, Wg, bg)
print("gen_z max rel err vs torch:", float((y - ref).abs().max() / ref.abs().max()))

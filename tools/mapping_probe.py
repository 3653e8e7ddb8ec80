"""Time the mapping network (8 x [10000 x 512 x 512] f32-MFMA layers) and gen_z-shaped Linear on the device."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ganspace_amd import ops
dev = torch.device("cuda", 0)
W, b = bench.make_mapping_weights(dev)
z = torch.randn(10000, 512, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): w = ops.mapping_forward(z, W, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    fl = 8 * 2 * 10000 * 512 * 512
    print(f"mapping 10000x512, 8 layers: {dt*1e6:.0f} us  ({fl/dt/1e12:.1f} TF/s, {dt/8*1e6:.1f} us per layer)")
x = torch.randn(2000, 256, device=dev); Wg = torch.randn(32768, 256, device=dev) * 0.05; bg = torch.zeros(32768, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): y = ops.linear_forward(x, Wg, bg)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"gen_z 2000x256 -> 32768: {dt*1e6:.0f} us  ({2*2000*256*32768/dt/1e12:.1f} TF/s)")
ref = torch.nn.functional.linear(x, Wg, bg)
print("gen_z max rel err vs torch:", float((y - ref).abs().max() / ref.abs().max()))

import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganspace_amd import _lib
from ganspace_amd.estimators import IPCAEstimator
lib = _lib.load()
dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
X = torch.randn(rows, d, device=dev)
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
est = IPCAEstimator(min(80, d), "exact", precision=prec); est.transformer._ensure(d)
ms = C.c_float(0); rt = C.c_int64(0)
for rep in range(3):
    _lib.check(lib.gs_gram_kernel_time(est.transformer._h, C.c_void_p(X.data_ptr()), rows, d, 50,
               C.cast(C.byref(ms), C.c_void_p), C.cast(C.byref(rt), C.c_void_p), _lib.current_stream_ptr()))
    fl = rt.value * d * (d + 1)
    print(f"{prec} rows={rt.value} d={d} wgs={os.environ.get('GS_GRAM_TARGET_WGS','512')} gram_partial {ms.value*1e3:.1f} us  useful {fl/ms.value/1e9:.1f} TF/s  {rt.value*d*4/ms.value/1e6:.0f} GB/s")
# whole update (partial + fold) timing
torch.cuda.synchronize()
import time
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(100): est.fit_partial(X)
    torch.cuda.synchronize()
    print(f"update (partial+fold) {(time.perf_counter()-t0)/100*1e6:.1f} us/block")

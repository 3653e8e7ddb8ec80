"""Same-box A/B of estimator builds: the cfg2 faithful job (100 blocks, no per-block synchronisation) and the exact-mode job's
finalize through the RAW C ABI of the library given on the command line (any build that exports gs_ipca_create / update /
update_resident / finalize - rounds 3-6), on blocks produced by the current package.
    python tools/estimator_ab_probe.py <path/to/libganspace_hip.so> [blocks]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
dev = torch.device("cuda", 0)
lib = C.CDLL(sys.argv[1])
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lib.gs_ipca_create.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
lib.gs_ipca_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
lib.gs_ipca_update_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
lib.gs_ipca_finalize.argtypes = [C.c_void_p] * 9
lib.gs_ipca_destroy.argtypes = [C.c_void_p]
blocks, steps, _, _ = bench.make_blocks(-(-nb // 5) * 5, dev)
blocks = blocks[:nb]
k, d = 80, 512
comp = np.empty((k, d), np.float32); sv, ev, evr = (np.empty(k) for _ in range(3)); mean, var = np.empty(d), np.empty(d); n = C.c_int64()
p = lambda a: a.ctypes.data_as(C.c_void_p)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def job(mode, resident):
    h = C.c_void_p()
    assert lib.gs_ipca_create(d, k, mode, 0, 0, C.byref(h)) == 0
    # (allocate the handle's workspaces outside the timed region: one throw-away block on a second handle is not needed -
    #  gs_ipca_create allocates everything)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if resident:
        for s in steps[:nb // 5]:
            assert lib.gs_ipca_update_resident(h, C.c_void_p(s.data_ptr()), s.shape[0], s.stride(0), stream) == 0
    else:
        for b in blocks:
            assert lib.gs_ipca_update(h, C.c_void_p(b.data_ptr()), b.shape[0], b.stride(0), stream) == 0
    torch.cuda.synchronize(); t1 = time.perf_counter()
    assert lib.gs_ipca_finalize(h, p(comp), p(sv), p(mean), p(var), p(ev), p(evr), C.cast(C.byref(n), C.c_void_p), stream) == 0
    torch.cuda.synchronize(); t2 = time.perf_counter()
    lib.gs_ipca_destroy(h)
    return t2 - t0, t2 - t1
for mode, name, resident in ((1, "faithful", False), (0, "exact", True)):
    job(mode, resident)
    runs = [job(mode, resident) for _ in range(7)]
    tot = min(r[0] for r in runs); fin = min(r[1] for r in runs)
    print(f"{os.path.basename(os.path.dirname(sys.argv[1])):12s} {name:8s} {nb} blocks: job {tot*1e3:7.3f} ms = {tot/nb*1e3:.4f} ms per block; finalize {fin*1e3:.3f} ms", flush=True)

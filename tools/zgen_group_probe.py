"""gs_zgen_device: duration of ONE launch against the number of streams in it (one workgroup of four waves per seed; 182
VGPRs per wave leave room for two workgroups per CU).  python tools/zgen_group_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ganspace_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
n, dim = 10000, 512
count = n * dim
# measurement build (GANSPACE_HIP_LIB=.../lib_measure/...): GS_ZGEN_GROUP_BLOCKS = blocks of 624 draws per pass of log / sqrt
measure = "lib_measure" in os.environ.get("GANSPACE_HIP_LIB", "")
variants = ["4", "2", "1"] if measure else [None]
for kind, cnt, grp in [(0, count, g) for g in variants] + [(1, 2000 * 128, None)]:
    if grp is not None:
        os.environ["GS_ZGEN_GROUP_BLOCKS"] = grp
        print(f"--- GS_ZGEN_GROUP_BLOCKS={grp}")
    for nseeds in (101, 256, 512, 801, 1024, 2048):
        if nseeds * cnt * 4 > 60e9:
            continue
        seeds = torch.from_numpy(np.random.RandomState(1).randint(0, 2**31 - 1, size=nseeds).astype(np.int32)).to(dev)
        buf = torch.empty((nseeds, cnt), dtype=torch.float32, device=dev)
        la, lm = (float.fromhex("-0x1.e43f625df3b24p+1"), float.fromhex("-0x1.7d7bfd8ad78c5p-5")) if kind else (0.0, 0.0)
        best = None
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _lib.check(lib.gs_zgen_device(C.c_void_p(seeds.data_ptr()), nseeds, cnt, C.c_void_p(buf.data_ptr()), cnt, kind, la, lm,
                                          1.0, _lib.current_stream_ptr()))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(f"kind {kind} count {cnt}: {nseeds:5d} seeds in one launch: {best*1e3:7.2f} ms  = {best/nseeds*1e3:.3f} ms per seed, "
              f"{nseeds*cnt/best/1e9:.1f} G values/s", flush=True)
        del buf

# few, long streams: every stream on one workgroup against streams cut into segments (gs_zgen_device_segmented)
from ganspace_amd import _zgen
for nseeds, rows in ((101, 10000), (1, 5000), (16, 10000), (400, 10000)):
    seeds = list(range(1000, 1000 + nseeds))
    out = torch.empty((nseeds, rows, 512), dtype=torch.float32, device=dev)
    for mode in ("0", "1"):
        os.environ["GANSPACE_ZGEN_SEGMENTS"] = mode
        best = None
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in _zgen.device_groups("stylegan", seeds, rows, 512, dev, out=out): pass
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(f"{nseeds:4d} streams of {rows} x 512 normals, segments {'on ' if mode == '1' else 'off'} "
              f"({_zgen.plan_segments(rows * 512) if mode == '1' else 1:2d} per stream): {best*1e3:7.2f} ms", flush=True)
    del out

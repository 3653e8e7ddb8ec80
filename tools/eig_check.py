"""Timing probe of the top-k eigensolver pieces on the GPU box (not a test; tests/test_gpu_topk.py checks results).

    python tools/eig_check.py        -> per-piece milliseconds: chol_inv, jacobi_small, exact finalize, faithful block
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ganspace_amd import _lib, ops
from ganspace_amd.estimators import IPCAEstimator

lib = _lib.load()
dev = torch.device("cuda", 0)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2]


rs = np.random.RandomState(0)
for p in (128, 96, 64):
    Q, _ = np.linalg.qr(rs.standard_normal((p, p)))
    H = torch.from_numpy((Q * np.logspace(0, -4, p)) @ Q.T).to(dev)
    Yt = torch.from_numpy(rs.standard_normal((512, p))).to(dev)
    print(f"cholqr n=512 p={p} (incl. workspace alloc): min/median ms", timeit(lambda: ops.cholqr(Yt)), flush=True)
    th, U, sw, lim = ops.jacobi_small(H)
    print(f"jacobi_small dense p={p}: sweeps={sw}", timeit(lambda: ops.jacobi_small(H)), flush=True)
    Bn = torch.from_numpy(np.diag(np.logspace(0, -3, p)) + 1e-7 * np.ones((p, p))).to(dev)
    th, U, sw, lim = ops.jacobi_small(Bn)
    print(f"jacobi_small near-diagonal p={p}: sweeps={sw}", timeit(lambda: ops.jacobi_small(Bn)), flush=True)

import bench
blocks, _, _ = bench.make_blocks(15, dev)
for mode in ("exact", "faithful"):
    for rep in range(3):
        est = IPCAEstimator(80, mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            est.fit_partial(blocks[i])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        est.get_components()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        h = est.transformer._h
        print(f"{mode}: 10 updates {1e3*(t1-t0):.2f} ms, finalize {1e3*(t2-t1):.2f} ms, "
              f"products={lib.gs_ipca_last_mults(h)} sweeps={lib.gs_ipca_last_sweeps(h)}", flush=True)

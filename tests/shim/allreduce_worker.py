"""Worker of tests/test_gpu_collective_shim.py: P "ranks" (host threads, one estimator handle each, all on the box's one
GPU) meet in ``gs_ipca_allreduce`` through the in-process RCCL stand-in ``tests/shim/libfake_rccl.so``.

The stand-in must be in the process (RTLD_GLOBAL) BEFORE the library resolves ncclAllReduce / ncclAllGather /
ncclCommCount with dlsym - hence its own process, and the load order below.

    python tests/shim/allreduce_worker.py <mode: exact|faithful> <P> <empty_rank or -1> <out.npz>
"""
import ctypes as C
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
shim = C.CDLL(os.path.join(HERE, "libfake_rccl.so"), mode=C.RTLD_GLOBAL)
shim.fake_rccl_group_create.restype = C.c_void_p
shim.fake_rccl_comm_create.restype = C.c_void_p
shim.fake_rccl_comm_create.argtypes = [C.c_void_p, C.c_int]
shim.fake_rccl_calls.argtypes = [C.c_void_p, C.c_int]

import numpy as np
import torch

from ganspace_amd import _lib
from ganspace_amd.estimators import IPCAEstimator


def main():
    mode, P, empty, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    d, k, rows, blocks = 96, 8, 400, 3
    rs = np.random.RandomState(11)
    A = rs.standard_normal((24, d)) * (1.25 ** -np.arange(24))[:, None]
    shards = []
    for r in range(P):
        xs = [(rs.standard_normal((rows, 24)) @ A + 0.05 * rs.standard_normal((rows, d)) + 0.3 * (r + 1)).astype(np.float32)
              for _ in range(blocks)]
        shards.append([] if r == empty else xs)
    ests, comms, streams = [], [], []
    group = C.c_void_p(shim.fake_rccl_group_create(P))
    for r in range(P):
        est = IPCAEstimator(k, mode)
        est.transformer._ensure(d)                      # a rank without samples still owns a handle
        for X in shards[r]:
            est.fit_partial(torch.from_numpy(X).to(dev))
        ests.append(est)
        comms.append(C.c_void_p(shim.fake_rccl_comm_create(group, r)))
        streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()
    rcs = [None] * P

    def run(r):
        with torch.cuda.device(dev):
            rcs[r] = lib.gs_ipca_allreduce(ests[r].transformer._h, comms[r], C.c_void_p(streams[r].cuda_stream))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert all(rc == 0 for rc in rcs), (rcs, lib.gs_last_error())
    res = {}
    for r in range(P):
        tr = ests[r].transformer
        tr._cache = None
        res[f"comp{r}"] = np.array(tr.components_)
        res[f"sv{r}"] = np.array(tr.singular_values_)
        res[f"mean{r}"] = np.array(tr.mean_)
        res[f"n{r}"] = np.int64(tr.n_samples_seen_)
    res["allreduce_calls"] = np.int64(shim.fake_rccl_calls(group, 0))
    res["allgather_calls"] = np.int64(shim.fake_rccl_calls(group, 1))
    flat = [X for xs in shards for X in xs]
    res["all_rows"] = np.concatenate(flat, axis=0)
    res["shard_rows"] = np.array([sum(len(X) for X in xs) for xs in shards], dtype=np.int64)
    np.savez(out, **res)
    print("OK", flush=True)


if __name__ == "__main__":
    main()

"""Host-side latent (z) generation, bit-identical to the reference's per-batch seeding, in parallel.

The reference draws one seed per mini-batch from the global legacy NumPy stream and generates the batch
from a private ``RandomState(seed)`` (``models/wrappers.py:167-174`` for StyleGAN2,
``models/biggan/.../utils.py:21-33`` for BigGAN).  MT19937 + the polar Gaussian are serial per seed
(88 k samples/s/core for 512-d z), i.e. 11 s for n = 1e6 - three orders of magnitude more
than the PCA on the GPU.  The batches are independent once the seed list is drawn.  Both latent kinds come from the
library's native thread pool (``NativeNormalStream`` over ``gs_zgen_*``, csrc/gs_zgen.hip: no interpreter start-up,
pinned ring buffers): StyleGAN's plain normals and, since round 4, BigGAN's ``scipy.stats.truncnorm`` latents (SciPy's
inverse-CDF chain restated in C++).  The worker *subprocesses* of round 2 (``python -m ganspace_amd._zgen``: NumPy /
SciPy only, writing into a shared memory-mapped array) remain behind ``GANSPACE_ZGEN_WORKERS`` as the SciPy-evaluated
cross-check.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np


def stylegan_z(seed: int, n: int, dim: int = 512) -> np.ndarray:
    rng = np.random.RandomState(seed)
    return rng.standard_normal(dim * n).reshape(n, dim).astype(np.float32)


def biggan_z(seed: int, n: int, dim: int = 128, truncation: float = 1.0) -> np.ndarray:
    from scipy.stats import truncnorm
    state = np.random.RandomState(seed)
    values = truncnorm.rvs(-2, 2, size=(n, dim), random_state=state).astype(np.float32)
    return truncation * values


def _one(kind, seed, n, dim, truncation):
    return stylegan_z(seed, n, dim) if kind == "stylegan" else biggan_z(seed, n, dim, truncation)


# (n_slots, count, pinned) -> [storage, ctypes pointers, in use]: pinning host memory is not free, so ONE ring is kept
# between streams.  A ring belongs to one stream at a time: a second stream that is opened while the first is alive
# (two generators zipped, a generator that was not exhausted) gets a private ring instead of the other's slots.
_RING_CACHE = {}

# log Phi(-2) and log(Phi(2) - Phi(-2)) as SciPy computes them (scipy.special.log_ndtr(-2.0),
# np.log1p(-ndtr(-2.0) - ndtr(-2.0))): the interval of BigGAN's truncated_noise_sample
TRUNCNORM_M2_P2 = (float.fromhex("-0x1.e43f625df3b24p+1"), float.fromhex("-0x1.7d7bfd8ad78c5p-5"))


class NativeNormalStream:
    """The StyleGAN z batches for ``seeds`` from the library's thread pool (``gs_zgen_*``: MT19937 + NumPy's legacy
    polar Gaussian restated in C++, bit-identical to ``RandomState(seed).standard_normal``), written into a ring of
    batch buffers - pinned host memory when a HIP device is present, so that the consumer can start an asynchronous
    H2D copy straight out of the slot.

    Iterating yields ``(index, batch)`` with ``batch`` a ``[n, dim]`` float32 view of a ring slot (a torch tensor if
    the ring is pinned, else a NumPy array).  A slot is recycled once ``release(index + 1)`` has been called - the
    consumer calls it when it is done with batch ``index`` (after the event of its H2D copy); the iterator itself
    never releases anything it has handed out."""

    def __init__(self, seeds, n: int, dim: int, threads=None, pinned=None, kind="stylegan", truncation=1.0):
        """``kind="biggan"``: ``truncation * truncnorm.rvs(-2, 2, size=(n, dim), random_state=RandomState(seed))`` in
        float32 instead of standard normals (``gs_zgen_start_truncnorm``)."""
        import ctypes as C
        from . import _lib
        self._lib = _lib.load()
        self._C = C
        self.seeds = np.asarray([int(s) for s in seeds], dtype=np.uint32)
        self.n, self.dim = int(n), int(dim)
        nb = len(self.seeds)
        env = os.environ.get("GANSPACE_ZGEN_THREADS")
        if threads is None:
            # (ranks of one node share its cores: torch.distributed.run exports LOCAL_WORLD_SIZE)
            local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
            threads = int(env) if env else min(64, max(1, (os.cpu_count() or 1) // local_world))
        threads = max(1, min(int(threads), max(nb, 1)))
        n_slots = min(max(nb, 1), threads + 4)
        if pinned is None:
            try:
                import torch
                pinned = torch.cuda.is_available()
            except Exception:
                pinned = False
        count = self.n * self.dim
        key = (n_slots, count, bool(pinned))

        def new_ring():
            if pinned:
                import torch
                storage = torch.empty((n_slots, self.n, self.dim), dtype=torch.float32, pin_memory=True)
                base = storage.data_ptr()
            else:
                storage = np.empty((n_slots, self.n, self.dim), dtype=np.float32)
                base = storage.ctypes.data
            return [storage, (C.c_void_p * n_slots)(*[base + i * count * 4 for i in range(n_slots)]), True]

        entry = _RING_CACHE.get(key)
        if entry is not None and not entry[2]:
            entry[2] = True                        # the cached ring is free: take it
        elif entry is not None or any(e[2] for e in _RING_CACHE.values()):
            entry = new_ring()                     # another live stream owns the cached ring: a private one, not cached
        else:
            entry = new_ring()
            _RING_CACHE.clear()                    # one idle ring at a time: a different shape replaces the old buffers
            _RING_CACHE[key] = entry
        self._ring = entry
        self._storage, self._ptrs = entry[0], entry[1]
        self._n_slots = n_slots
        self._h = C.c_void_p()
        try:
            if kind == "biggan":
                la, lm = TRUNCNORM_M2_P2
                _lib.check(self._lib.gs_zgen_start_truncnorm(self.seeds.ctypes.data_as(C.c_void_p), nb, count,
                                                             C.cast(self._ptrs, C.c_void_p), n_slots, threads, la, lm,
                                                             float(truncation), C.byref(self._h)))
            elif kind == "stylegan":
                _lib.check(self._lib.gs_zgen_start(self.seeds.ctypes.data_as(C.c_void_p), nb, count,
                                                   C.cast(self._ptrs, C.c_void_p), n_slots, threads, C.byref(self._h)))
            else:
                raise ValueError(f"unknown latent kind {kind!r}")
        except Exception:
            entry[2] = False
            raise
        self.threads = threads

    def __len__(self):
        return len(self.seeds)

    def __iter__(self):
        from . import _lib
        C = self._C
        for i in range(len(self.seeds)):
            slot = C.c_void_p()
            _lib.check(self._lib.gs_zgen_wait(self._h, i, C.byref(slot)))
            yield i, self._storage[i % self._n_slots]

    def release(self, upto: int):
        if self._h:
            self._lib.gs_zgen_release(self._h, int(upto))

    def close(self):
        if self._h:
            self._lib.gs_zgen_finish(self._h)      # joins the pool: nobody writes into the ring after this
            self._h = None
            self._ring[2] = False                  # (a private ring simply goes away with the stream)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_generation_enabled(device) -> bool:
    """The device generator (``gs_zgen_device``) serves every HIP run unless ``GANSPACE_ZGEN=host`` asks for the host thread
    pool (A/B timing, or bit-for-bit glibc rows)."""
    import torch
    return torch.device(device).type == "cuda" and os.environ.get("GANSPACE_ZGEN", "device") != "host"


ZGEN_STAGING_BYTES = 12 << 30     # device staging of one gs_zgen_device launch (device_groups), and never more than a
ZGEN_STAGING_FRACTION = 4         # quarter of the free device memory
# streams per launch: one workgroup (four waves, 54 VGPRs, 20 KB of LDS) per stream, up to eight workgroups per CU - measured
# per launch of 10 000 x 512 normals (tools/zgen_group_probe.py): 256 streams 21.4 ms, 512 26.0 ms, 1024 38.4 ms, 2048 66.5 ms
# (rounds 5-6, 182 VGPRs, two workgroups per CU: 22.9 / 30.5 / 58.6 / 113.5 ms)
ZGEN_GROUP = 2048


# One stream on many workgroups (gs_zgen_device_segmented): MT19937 jump-ahead polynomials for offsets of i * 2 048 blocks of
# 624 draws (tools/make_mt_jump.py wrote the file and checked every polynomial against NumPy).  A launch of few, long streams
# - cfg2: 101 streams of 5.12 M normals, 21 ms on 101 of the 256 CUs whatever else is idle - is cut into segments that run
# side by side; launches of more than SEGMENT_MAX_STREAMS streams fill the chip as they are.
JUMP_BLOCK_LEN = 2048
SEGMENT_MAX_STREAMS = 512
VALUES_PER_BLOCK = 2 * 156 * np.pi / 4        # 156 candidates of the polar method per block, accepted with probability pi / 4
_JUMP_POLYS = {}


def plan_segments(count: int) -> int:
    """Segments of JUMP_BLOCK_LEN blocks that hold ``count`` normals of one stream with 0.4 % + 2 blocks to spare (the number
    of accepted candidates of B blocks has a standard deviation of sqrt(26 B) pairs: 6 blocks' worth for the 20 900 blocks
    of a 10 000 x 512 mini-batch); 1 = not worth cutting, 0 = longer than the polynomial file reaches."""
    blocks = int(np.ceil(count / VALUES_PER_BLOCK * 1.004)) + 2
    segments = -(-blocks // JUMP_BLOCK_LEN)
    return segments if segments <= jump_polys_host().shape[0] + 1 else 0


def jump_polys_host() -> np.ndarray:
    if "host" not in _JUMP_POLYS:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"mt19937_jump_L{JUMP_BLOCK_LEN}.npz")
        with np.load(path) as f:
            assert int(f["block_len"]) == JUMP_BLOCK_LEN
            _JUMP_POLYS["host"] = np.ascontiguousarray(f["polys"], dtype=np.uint32)
    return _JUMP_POLYS["host"]


def _jump_polys_device(device):
    import torch
    key = str(torch.device(device))
    if key not in _JUMP_POLYS:
        _JUMP_POLYS[key] = torch.from_numpy(jump_polys_host().view(np.int32)).to(device)
    return _JUMP_POLYS[key]


def segmented_generation_enabled() -> bool:
    """``GANSPACE_ZGEN_SEGMENTS=0`` keeps every stream on one workgroup (A/B timing)."""
    return os.environ.get("GANSPACE_ZGEN_SEGMENTS", "1") != "0"


def device_groups(kind: str, seeds, n: int, dim: int, device, truncation: float = 1.0, group: int = None, out=None):
    """Yield ``(lo, z)`` for ``seeds`` in order, ``z`` a ``[m, n, dim]`` float32 DEVICE tensor whose slice ``z[j]`` holds
    ``RandomState(seeds[lo + j]).standard_normal(n * dim)`` (``kind="stylegan"``) or BigGAN's
    ``truncation * truncnorm.rvs(-2, 2, size=(n, dim), random_state=RandomState(seed))`` - generated on the device, one
    workgroup (four waves) per seed, up to ``group`` seeds per launch.  A launch of up to 256 streams lasts as long as ONE
    stream (21 ms for 10 000 x 512 normals), 2 048 streams three times as long (up to eight workgroups per CU): the groups are
    as long as memory allows - ``out`` (``[len(seeds), n, dim]``, e.g. the resident latent array of a Z-space job: no staging
    at all) or a staging buffer of at most ``ZGEN_STAGING_BYTES`` / a quarter of the free device memory.  (Round 6 first
    capped the staging at 1 GiB: cfg4's 801 streams became 16 launches instead of 4, 0.35 s instead of 0.09 s.)  Nothing
    touches the host: no pinned ring, no H2D copy."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    seeds = np.asarray([int(s) for s in seeds], dtype=np.uint32)
    count = int(n) * int(dim)
    if kind == "biggan":
        knd, (la, lm), scale = 1, TRUNCNORM_M2_P2, float(truncation)
    elif kind == "stylegan":
        knd, (la, lm), scale = 0, (0.0, 0.0), 1.0
    else:
        raise ValueError(f"unknown latent kind {kind!r}")
    if len(seeds) == 0:
        return
    seeds_dev = torch.from_numpy(seeds.view(np.int32).copy()).to(device)
    group = max(1, min(int(group or ZGEN_GROUP), len(seeds)))
    if out is None:
        budget = ZGEN_STAGING_BYTES
        try:
            budget = min(budget, torch.cuda.mem_get_info(device)[0] // ZGEN_STAGING_FRACTION)
        except Exception:
            pass
        group = max(1, min(group, budget // max(1, 4 * count)))
    else:
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == len(seeds) * count and out.is_cuda
        out = out.view(len(seeds), int(n), int(dim))
    stream = _lib.current_stream_ptr()
    segments = plan_segments(count) if (knd == 0 and segmented_generation_enabled()) else 1
    for lo in range(0, len(seeds), group):
        m = min(group, len(seeds) - lo)
        buf = out[lo:lo + m] if out is not None else torch.empty((m, int(n), int(dim)), dtype=torch.float32, device=device)
        done = False
        if segments >= 2 and m <= SEGMENT_MAX_STREAMS:
            nbytes = C.c_int64(0)
            _lib.check(lib.gs_zgen_segmented_nbytes(m, segments, JUMP_BLOCK_LEN, C.cast(C.byref(nbytes), C.c_void_p)))
            try:
                room = torch.cuda.mem_get_info(device)[0] // 2
            except Exception:
                room = nbytes.value
            if nbytes.value <= room:
                scratch = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
                short = C.c_int(0)
                _lib.check(lib.gs_zgen_device_segmented(C.c_void_p(seeds_dev.data_ptr() + 4 * lo), m, count,
                                                        C.c_void_p(buf.data_ptr()), count,
                                                        C.c_void_p(_jump_polys_device(device).data_ptr()), JUMP_BLOCK_LEN,
                                                        segments, C.c_void_p(scratch.data_ptr()), nbytes.value,
                                                        C.cast(C.byref(short), C.c_void_p), stream))
                done = short.value == 0          # a stream short of accepted candidates (a 7-sigma event): serial path below
                del scratch
        if not done:
            _lib.check(lib.gs_zgen_device(C.c_void_p(seeds_dev.data_ptr() + 4 * lo), m, count, C.c_void_p(buf.data_ptr()),
                                          count, knd, la, lm, scale, stream))
        yield lo, buf


def device_batches(kind: str, seeds, n: int, dim: int, device, truncation: float = 1.0, group: int = None):
    """:func:`device_groups` one stream at a time: yields ``(index, z)`` with ``z`` the ``[n, dim]`` batch of
    ``seeds[index]``."""
    for lo, buf in device_groups(kind, seeds, n, dim, device, truncation, group):
        for j in range(buf.shape[0]):
            yield lo + j, buf[j]


def generate(kind: str, seeds, n: int, dim: int, truncation: float = 1.0, workers=None):
    """Yield the z batches for ``seeds`` in order (NumPy arrays the caller may keep).  StyleGAN batches come from the
    library's native generator; BigGAN's ``truncnorm.rvs`` batches from worker subprocesses when the job is large
    enough to pay for their start-up (or when ``GANSPACE_ZGEN_WORKERS`` forces a worker count)."""
    seeds = [int(s) for s in seeds]
    if kind in ("stylegan", "biggan") and len(seeds) > 0 and os.environ.get("GANSPACE_ZGEN_WORKERS") is None:
        stream = NativeNormalStream(seeds, n, dim, pinned=False, kind=kind, truncation=truncation)
        try:
            for i, z in stream:
                out = np.array(z, copy=True)
                stream.release(i + 1)
                yield out
        finally:
            stream.close()
        return
    env = os.environ.get("GANSPACE_ZGEN_WORKERS")
    if workers is None:
        workers = int(env) if env else min(32, os.cpu_count() or 1)
    workers = min(workers, len(seeds))
    elements = len(seeds) * n * dim
    cost = 1.0 if kind == "stylegan" else 40.0      # truncnorm.rvs is far slower per element
    if workers <= 1 or len(seeds) < 4 or (env is None and elements * cost < 2e8):
        for s in seeds:
            yield _one(kind, s, n, dim, truncation)
        return

    shm_dir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=shm_dir, prefix="ganspace_z_") as td:
        data_path, done_path = os.path.join(td, "z.f32"), os.path.join(td, "done.u8")
        data = np.lib.format.open_memmap(data_path, mode="w+", dtype=np.float32, shape=(len(seeds), n, dim))
        done = np.lib.format.open_memmap(done_path, mode="w+", dtype=np.uint8, shape=(len(seeds),))
        done[:] = 0
        done.flush()
        pkg_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env_child = dict(os.environ, PYTHONPATH=pkg_root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs = []
        for w in range(workers):
            job = dict(kind=kind, n=n, dim=dim, truncation=truncation, data=data_path, done=done_path,
                       items=[(i, seeds[i]) for i in range(w, len(seeds), workers)])
            p = subprocess.Popen([sys.executable, "-m", "ganspace_amd._zgen"], stdin=subprocess.PIPE, env=env_child)
            p.stdin.write(json.dumps(job).encode())
            p.stdin.close()
            procs.append(p)
        # flag protocol (one byte per batch): 0 untouched, 2 claimed by a worker, 1 ready in `data`, 3 taken by the
        # parent.  While the workers are still starting up (python + numpy import, ~1-2 s on a cold box) the parent
        # generates the batches it needs itself instead of waiting; a worker skips what the parent took.  A lost
        # race only means a batch is generated twice - the result is a pure function of its seed.
        try:
            for i in range(len(seeds)):
                if done[i] == 0:
                    done[i] = 3
                    yield _one(kind, seeds[i], n, dim, truncation)
                    continue
                while done[i] != 1:
                    if any(p.poll() not in (None, 0) for p in procs):
                        raise RuntimeError("z-generation worker failed")
                    time.sleep(0.0005)
                yield data[i]          # a view into the shared map: consume (copy) it before the next batch is requested
        finally:
            for p in procs:
                if p.poll() is None:
                    p.wait(timeout=60)
            del data, done


def _worker_main():
    job = json.loads(sys.stdin.buffer.read().decode())
    data = np.load(job["data"], mmap_mode="r+")
    done = np.load(job["done"], mmap_mode="r+")
    # No msync: parent and workers map the same file MAP_SHARED, which is coherent without it (and on a multi-GB map
    # every flush() walked the whole mapping).  The flag is written after the data; x86 keeps the stores in order.
    for i, seed in job["items"]:
        if done[i] != 0:            # the parent took this batch while we were starting up
            continue
        done[i] = 2
        data[i] = _one(job["kind"], seed, job["n"], job["dim"], job["truncation"])
        done[i] = 1


if __name__ == "__main__":
    _worker_main()

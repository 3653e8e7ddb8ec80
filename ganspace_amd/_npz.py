"""``np.savez_compressed`` with the members deflated in parallel.

The component file of a job (decomposition.py:331-341) is an ``.npz``: a zip archive of ``.npy`` members, deflated.  Float32
components barely compress (10.5 MB -> 9.7 MB for cfg3's 80 x 32 768), and zlib spends 20 MB/s on finding that out: 0.34 s
of a 2.0 s cfg3 job, 1.1 s of cfg5's, on one host core.  A deflate stream may be assembled from independently compressed
pieces - each piece ends on a byte boundary with ``Z_SYNC_FLUSH``, the last one with ``Z_FINISH`` (the construction of
pigz) - so the pieces go to a thread pool (zlib releases the GIL) and the archive is written by hand: local headers,
central directory, end record (PKWARE APPNOTE 4.3; no zip64: members of 2 GiB or more go through NumPy's own writer).
Same keys, same arrays, readable by ``np.load`` like any other ``.npz``."""
import io
import os
import struct
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

PIECE_BYTES = 1 << 20
MAX_MEMBER_BYTES = (1 << 31) - 1


def _deflate_piece(data, last):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    return co.compress(data) + co.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH)


def _dos_time(t=None):
    tm = time.localtime(t)
    year = max(tm.tm_year, 1980)
    return (tm.tm_hour << 11) | (tm.tm_min << 5) | (tm.tm_sec // 2), ((year - 1980) << 9) | (tm.tm_mon << 5) | tm.tm_mday


def savez_compressed(path, threads=None, **arrays):
    """Drop-in for ``np.savez_compressed(path, **arrays)``."""
    path = os.fspath(path)
    if not path.endswith(".npz"):
        path += ".npz"
    members = []
    for name, value in arrays.items():
        buf = io.BytesIO()
        np.lib.format.write_array(buf, np.asanyarray(value), allow_pickle=False)
        raw = buf.getvalue()
        if len(raw) > MAX_MEMBER_BYTES:
            return np.savez_compressed(path, **arrays)
        members.append((name + ".npy", raw))
    threads = threads or min(16, os.cpu_count() or 1)
    dtime, ddate = _dos_time()
    with ThreadPoolExecutor(threads) as pool:
        jobs = []
        for fname, raw in members:
            pieces = [raw[i:i + PIECE_BYTES] for i in range(0, len(raw), PIECE_BYTES)] or [b""]
            jobs.append((fname, raw, [pool.submit(_deflate_piece, p, i == len(pieces) - 1) for i, p in enumerate(pieces)]))
        central = []
        with open(path, "wb") as f:
            for fname, raw, futures in jobs:
                data = b"".join(fu.result() for fu in futures)
                name = fname.encode("utf-8")
                crc = zlib.crc32(raw) & 0xFFFFFFFF
                offset = f.tell()
                if offset + len(data) > MAX_MEMBER_BYTES:
                    f.close()
                    return np.savez_compressed(path, **arrays)
                # local file header: signature, version needed 2.0, flags (bit 11: UTF-8 name), method 8 = deflate
                f.write(struct.pack("<IHHHHHIIIHH", 0x04034B50, 20, 0x800, 8, dtime, ddate, crc, len(data), len(raw),
                                    len(name), 0))
                f.write(name)
                f.write(data)
                central.append(struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 20, 20, 0x800, 8, dtime, ddate, crc, len(data),
                                           len(raw), len(name), 0, 0, 0, 0, 0, offset) + name)
            start = f.tell()
            for entry in central:
                f.write(entry)
            size = f.tell() - start
            f.write(struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, len(central), len(central), size, start, 0))

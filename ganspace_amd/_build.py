"""Build the gfx950 shared library in-tree with hipcc (no JIT cache, no CPU fallback).

    python -m ganspace_amd._build          # builds ganspace_amd/lib/libganspace_hip.so

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container; the
resulting .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.environ.get("GS_LIBDIR") or os.path.join(PKG, "lib")     # (GS_LIBDIR: side-by-side measurement builds)
LIBNAME = "libganspace_hip.so"
SOURCES = ["gs_collective.hip", "gs_gram.hip", "gs_eigh.hip", "gs_ipca.hip", "gs_linear.hip", "gs_smallside.hip", "gs_subspace.hip", "gs_topk.hip", "gs_gram_bf16.hip", "gs_gram_wide.hip", "gs_zgen.hip", "gs_rangefinder.hip"]


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    out = lib_path()
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "ganspace_hip.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every source whose object is older than it (or than any header), then link.
    ``GS_HIPCC_FLAGS`` appends flags (e.g. ``-DGS_GRAM_ABLATE_BUILD`` for the measurement variants); objects are
    rebuilt when the flags change."""
    os.makedirs(LIBDIR, exist_ok=True)
    out = lib_path()
    extra = os.environ.get("GS_HIPCC_FLAGS", "").split()
    stamp = os.path.join(LIBDIR, "flags.txt")
    old = open(stamp).read() if os.path.exists(stamp) else ""
    if old != " ".join(extra):
        force = True
    if not force and not needs_build():
        return out
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "ganspace_hip.h"))
    newest_header = max(os.path.getmtime(h) for h in headers)
    objs, jobs = [], []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            continue
        jobs.append([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", *extra,
                     "-I", os.path.join(ROOT, "include"), "-I", CSRC, path, "-o", obj])

    def compile_one(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:      # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, jobs))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(" ".join(extra))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

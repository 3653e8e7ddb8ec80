"""Build the gfx950 shared library in-tree with hipcc (no JIT cache, no CPU fallback).

    python -m ganspace_amd._build          # builds ganspace_amd/lib/libganspace_hip.so

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container; the
resulting .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.environ.get("GS_LIBDIR") or os.path.join(PKG, "lib")     # (GS_LIBDIR: side-by-side measurement builds)
LIBNAME = "libganspace_hip.so"
SOURCES = ["gs_collective.hip", "gs_gram.hip", "gs_eigh.hip", "gs_ipca.hip", "gs_linear.hip", "gs_smallside.hip", "gs_subspace.hip", "gs_topk.hip", "gs_gram_bf16.hip", "gs_gram_wide.hip", "gs_zgen.hip", "gs_zgen_device.hip", "gs_rangefinder.hip", "gs_dense64.hip", "gs_tridiag.hip", "gs_gemm_blocked.hip"]


MEASURE_LIBDIR = os.path.join(PKG, "lib_measure")    # -DGS_MEASURE_BUILD: the A/B switches (gs_knob) read the environment


# kernels whose build fails if the register allocator gives them scratch (mangled-name substrings)
NO_SCRATCH = re.compile(r"gemm_blocked|block_rows|gram_|rowgram|tn_gemm|tn_rows|linear_act|project_rows|ss_build|mm64_|chol_inv|jacobi_lds|tridiag")


def _kernel_usage(remarks: str) -> dict:
    """{kernel: {vgprs, scratch, vgpr_spill, sgpr_spill, lds}} from -Rpass-analysis=kernel-resource-usage remarks."""
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def lib_path(measure: bool = False) -> str:
    return os.path.join(MEASURE_LIBDIR if measure else LIBDIR, LIBNAME)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build(measure: bool = False) -> bool:
    out = lib_path(measure)
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "ganspace_hip.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = True, measure: bool = False) -> str:
    """Compile every source whose object is older than it (or than any header), then link.
    ``GS_HIPCC_FLAGS`` appends flags (e.g. ``-DGS_GRAM_ABLATE_BUILD`` for the measurement variants); objects are
    rebuilt when the flags change.  ``measure=True`` builds the side-by-side library ``lib_measure/`` with
    ``-DGS_MEASURE_BUILD``: the only build whose A/B switches (``gs_knob`` in csrc/gs_common.h) read the environment;
    ``GANSPACE_HIP_LIB`` selects it at load time (tools/, one GPU test)."""
    libdir = MEASURE_LIBDIR if measure else LIBDIR
    os.makedirs(libdir, exist_ok=True)
    out = lib_path(measure)
    extra = os.environ.get("GS_HIPCC_FLAGS", "").split() + (["-DGS_MEASURE_BUILD"] if measure else [])
    stamp = os.path.join(libdir, "flags.txt")
    old = open(stamp).read() if os.path.exists(stamp) else ""
    if old != " ".join(extra):
        force = True
    if not force and not needs_build(measure):
        return out
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "ganspace_hip.h"))
    newest_header = max(os.path.getmtime(h) for h in headers)
    objs, jobs = [], []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(libdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            continue
        jobs.append([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", *extra,
                     "-I", os.path.join(ROOT, "include"), "-I", CSRC, path, "-o", obj])

    def compile_one(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        # the compiler reports every kernel's registers / scratch (-Rpass-analysis=kernel-resource-usage): the hot kernels
        # must not touch scratch - the ones that count their VMEM operations (`s_waitcnt vmcnt(N)` pipelines of the
        # LDS-DMA Gram kernel) are WRONG, not slow, once a spill reload shifts the count - so a spill fails the build
        r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
        usage = _kernel_usage(r.stderr)
        rest = "\n".join(l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l)
        if rest.strip() and (verbose or r.returncode != 0):
            print(rest, file=sys.stderr, flush=True)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        bad = [(k, u) for k, u in usage.items()
               if NO_SCRATCH.search(k) and (u.get("scratch", 0) > 0 or u.get("vgpr_spill", 0) > 0)]
        if bad:
            raise RuntimeError("kernels that must not spill use scratch: " +
                               "; ".join(f"{k}: {u}" for k, u in bad))
        return usage

    if jobs:      # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            usages = list(pool.map(compile_one, jobs))
        import json
        usage_path = os.path.join(libdir, "kernel_usage.json")
        merged = {}
        if not force and os.path.exists(usage_path):      # (an incremental build recompiles only some translation units)
            try:
                merged = json.load(open(usage_path))
            except Exception:
                merged = {}
        for u in usages:
            merged.update(u)
        with open(usage_path, "w") as f:     # (read by tests/test_abi.py)
            json.dump(merged, f, indent=0, sort_keys=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(" ".join(extra))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--measure" in sys.argv:
        print(build(force="--force" in sys.argv, measure=True))

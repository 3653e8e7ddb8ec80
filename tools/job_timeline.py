#!/usr/bin/env python3
"""Kernel timeline of ONE headline job out of a `rocprofv3 --kernel-trace` run of `bench.py --no-extras`: every Gram /
fold kernel between two eigensolves (the projection kernel - Jacobi, or the tridiagonal eigenvector kernel since round 4 - marks the end of a job), with start, duration and the overlap of
the fold kernels (second stream) with the compute launches.  Usage: job_timeline.py <dir with *kernel_trace.csv> [job]"""
import csv, glob, os, sys

root = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[-44:]
ends = [i for i, r in enumerate(rows) if "jacobi" in name(r) or "tridiag_eigvec" in name(r)]   # (the projection step ends a job)
j1 = ends[which]
j0 = ends[which - 1] if len(ends) > 1 else -1
seg = rows[j0 + 1:j1 + 1]
first = next(i for i, r in enumerate(seg) if "wide" in name(r) or "glds" in name(r))
seg = seg[max(0, first - 2):]
t0 = int(seg[0]["Start_Timestamp"])
print("| start us | duration us | kernel | overlaps the compute launch |\n|---|---|---|---|")
comp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg if "wide" in name(r) or "glds" in name(r)]
for r in seg:
    n = name(r)
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if not any(k in n for k in ("wide", "glds", "fold", "colsum", "mean_from", "jacobi", "tridiag", "assemble")):
        continue
    ov = ""
    if "fold" in n:
        o = sum(max(0, min(e, ce) - max(s, cs)) for cs, ce in comp)
        ov = f"{o / 1e3:.1f} of {(e - s) / 1e3:.1f} us"
    print(f"| {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | `{n}` | {ov} |")
last = seg[-1]
print(f"\njob span (first kernel listed -> end of the projection kernel): {(int(last['End_Timestamp']) - t0) / 1e3:.1f} us")

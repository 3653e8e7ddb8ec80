#!/usr/bin/env python3
"""Condense rocprofv3 outputs under <dir> into a markdown summary (kernel stats + per-launch PMC
averages for our kernels).  FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes
for wide coalesced streaming reads on gfx950 (the counter tallies 128-B requests as 64 B)."""
import collections
import csv
import os
import sys

root = sys.argv[1]
PROBE_ROWS = int(os.environ.get("GS_PROBE_ROWS", "131072"))
print(f"# rocprofv3 summary ({root})\n")
ks = os.path.join(root, "bench_trace", "b_kernel_stats.csv")
if os.path.exists(ks):
    print("## kernel-trace --stats of `python bench.py --no-cpu-baseline`\n")
    print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
    for r in list(csv.DictReader(open(ks)))[:12]:
        name = r["Name"].split("(")[0].replace("void ", "")[-60:]
        print(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    print()
print(f"## PMC passes on `tools/gram_probe.py` ({PROBE_ROWS} x 512 float32 rows per launch), averages per launch\n")
print("| kernel | counter | avg per launch | launches |\n|---|---|---|---|")
for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_sq", "pmc_lds"):
    p = os.path.join(root, sub, "p_counter_collection.csv")
    if not os.path.exists(p):
        continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "gs::" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")[-48:]
        d[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[short]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        for c, x in v.items():
            avg = sum(x) / len(x)
            note = ""
            if c == "FETCH_SIZE":
                note = f" KiB -> x2 correction = {avg*2*1024/1e6:.2f} MB HBM read"
            if c == "WRITE_SIZE":
                note = f" KiB = {avg*1024/1e6:.2f} MB written"
            print(f"| `{k}` | {c} | {avg:.1f}{note} | {len(x)} |")

# machine-readable companion for bench.py's roofline.traffic (written next to the summary when asked)
if len(sys.argv) > 2:
    import json
    vals = {}
    for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_sq"):
        p = os.path.join(root, sub, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            if "gram_f32_wide_kernel" in r["Kernel_Name"] or "gram_partial_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if sub == "pmc_FETCH_SIZE":
                    acc["_duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for c, x in acc.items():
            vals[c] = (sum(x) / len(x), len(x))
    fetch = vals.get("FETCH_SIZE", (0, 0))
    write = vals.get("WRITE_SIZE", (0, 0))
    alg = PROBE_ROWS * 512 * 4
    rd, wr = int(fetch[0] * 2 * 1024), int(write[0] * 1024)
    out = {
        "kernel": "gram_f32_wide_kernel (d = 512, >= 20 000 rows) / gram_partial_kernel", "rows_per_launch": PROBE_ROWS,
        "launches": fetch[1],
        "FETCH_SIZE_KiB_raw": round(fetch[0], 1),
        "hbm_read_bytes_per_launch_corrected_x2": rd,
        "WRITE_SIZE_KiB_raw": round(write[0], 1), "hbm_write_bytes_per_launch": wr,
        "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": round((rd + wr) / alg, 3),
        "note": "rocprofv3 --pmc passes on tools/gram_probe.py (separate passes per counter group; the probe launches "
                "include the piggy-backed fold workgroups); FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section",
        "source": sys.argv[2],
        "traffic_breakdown": "%.1f MB read + %.1f MB written per launch against %.1f MB of X rows (algorithmic): the rows are "
                             "fetched once; the rest is the float32 partial-Gram slabs (written by this launch, read back by "
                             "the fold workgroups of the next one)" % (rd / 1e6, wr / 1e6, alg / 1e6),
    }
    if "_duration_us" in vals:
        out["avg_launch_us_under_profiler"] = round(vals["_duration_us"][0], 1)
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES"):
        if c in vals:
            out[c] = int(vals[c][0])
    # the file holds one section per precision (bench.py reads "f32" for `roofline.traffic`, "bf16" for `roofline_hbm`):
    # this script measures the f32 launch and leaves the other sections as they are
    path = os.path.join("profiles", "gram_pmc_latest.json")
    doc = {}
    try:
        doc = json.load(open(path))
        if "f32" not in doc and "bf16" not in doc:
            doc = {}
    except Exception:
        doc = {}
    doc["f32"] = out
    json.dump(doc, open(path, "w"), indent=1)

#!/usr/bin/env python3
"""Condense the round-4 rocprofv3 outputs (tools/r04_profiles.sh) into markdown + the machine-readable
profiles/gram_pmc_latest.json that bench.py reads for roofline.traffic.  FETCH_SIZE is doubled as
MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on gfx950."""
import collections, csv, glob, json, os, sys

root = sys.argv[1]


def stats(sub, top=14):
    fs = glob.glob(os.path.join(root, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not fs:
        return
    print(f"### `{sub}`\n\n| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
    for r in list(csv.DictReader(open(fs[0])))[:top]:
        name = r["Name"].split("(")[0].replace("void ", "")[-64:]
        print(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    print()


def pmc(sub, match=None):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gs::" not in k or (match and match not in k):
                continue
            short = k.split("(")[0].replace("void ", "")[-56:]
            out[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out[short]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


print(f"# rocprofv3 summary, round 4 ({root})\n")
print("## kernel-trace --stats\n")
for sub in ("bench_trace", "finalize_trace", "ss131_trace", "ss131f_trace", "ss32_trace"):
    stats(sub)
print("## PMC passes (averages per launch; FETCH_SIZE x 2 per the guide's gfx950 correction)\n")
print("| run | kernel | counter | avg per launch | launches |\n|---|---|---|---|---|")
latest = {}
for tag, rows in (("f32", 131072), ("bf16", 1000000), ("ss", None)):
    merged = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE", "sq"):
        for k, v in pmc(f"pmc_{tag}_{c}").items():
            for cn, x in v.items():
                merged[k][cn] = (sum(x) / len(x), len(x))
    for k, v in merged.items():
        for cn, (avg, n) in sorted(v.items()):
            note = ""
            if cn == "FETCH_SIZE":
                note = f" KiB -> x2 = {avg*2*1024/1e6:.1f} MB read"
            if cn == "WRITE_SIZE":
                note = f" KiB = {avg*1024/1e6:.1f} MB written"
            print(f"| {tag} | `{k}` | {cn} | {avg:.1f}{note} | {n} |")
        if rows and ("wide" in k or "glds" in k) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            rd, wr = v["FETCH_SIZE"][0] * 2 * 1024, v["WRITE_SIZE"][0] * 1024
            latest[tag] = {"kernel": k, "rows_per_launch": rows, "hbm_read_bytes_per_launch_corrected_x2": int(rd),
                           "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
                           "algorithmic_bytes_per_launch": rows * 2048,
                           "traffic_over_algorithmic": round((rd + wr) / (rows * 2048), 3),
                           "avg_launch_us_under_profiler": round(v["duration_us"][0], 1)}
            for cn in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
                if cn in v:
                    latest[tag][cn] = int(v[cn][0])
if latest:
    with open(os.path.join(root, "gram_pmc_r04.json"), "w") as f:
        json.dump(latest, f, indent=1)

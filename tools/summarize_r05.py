#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of a round-5 gpurun call (tools/r05_run*.sh) into markdown: per-kernel stats of
every `--kernel-trace --stats` directory and per-kernel averages of every `--pmc` directory (FETCH_SIZE doubled as
MI355X_MICROARCH.md's HBM section prescribes for wide coalesced reads on gfx950)."""
import collections, csv, glob, os, sys

root = sys.argv[1]
print(f"# rocprofv3 summary ({root})\n")
for sub in sorted(os.listdir(root)):
    path = os.path.join(root, sub)
    if not os.path.isdir(path):
        continue
    fs = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
    if fs:
        print(f"### `{sub}` kernel stats\n\n| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
        for r in list(csv.DictReader(open(fs[0])))[:16]:
            name = r["Name"].split("(")[0].replace("void ", "")[-64:]
            print(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
        print()
    cs = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if cs:
        out = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in cs:
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "gs::" not in k:
                    continue
                short = k.split("(")[0].replace("void ", "")[-56:]
                out[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                out[short]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print(f"### `{sub}` counters (average per launch)\n\n| kernel | launches | " )
        rows = sorted(out.items(), key=lambda kv: -sum(kv[1]["duration_us"]))[:10]
        for kname, cnt in rows:
            n = len(cnt["duration_us"]) // max(1, len([c for c in cnt if c != "duration_us"]))
            parts = []
            for c, v in sorted(cnt.items()):
                avg = sum(v) / len(v)
                if c == "FETCH_SIZE":
                    parts.append(f"FETCH_SIZE x2 = {2 * avg / 1e6:.1f} GB" if avg > 1e3 else f"FETCH_SIZE x2 = {2*avg:.1f} KB")
                elif c == "WRITE_SIZE":
                    parts.append(f"WRITE_SIZE = {avg / 1e6:.2f} GB(KB units)")
                elif c == "duration_us":
                    parts.append(f"dur {avg:.1f} us")
                else:
                    parts.append(f"{c} {avg:.4g}")
            print(f"- `{kname}` x{n}: " + "; ".join(parts))
        print()

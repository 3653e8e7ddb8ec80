#!/usr/bin/env python3
"""Per-kernel averages of one rocprofv3 --pmc pass:  python tools/pmc_table.py <dir with *counter_collection.csv> [name filter]
FETCH_SIZE is reported raw (KiB) and x2-corrected in MB (MI355X_MICROARCH.md, HBM section), WRITE_SIZE in MB."""
import collections, csv, glob, os, sys
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "gs::"
d = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if flt not in k:
            continue
        short = k.split("(")[0].replace("void ", "")[-56:]
        d[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[short]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | counter | avg per launch | launches |\n|---|---|---|---|")
for k, v in sorted(d.items()):
    for c, x in v.items():
        avg = sum(x) / len(x)
        note = ""
        if c == "FETCH_SIZE":
            note = f" KiB (x2 -> {avg * 2 * 1024 / 1e6:.2f} MB read)"
        elif c == "WRITE_SIZE":
            note = f" KiB ({avg * 1024 / 1e6:.2f} MB written)"
        print(f"| `{k}` | {c} | {avg:.1f}{note} | {len(x)} |")

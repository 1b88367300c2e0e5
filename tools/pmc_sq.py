#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for p in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].replace("void adn::(anonymous namespace)::", "").split("(")[0][:60]
        if flt and flt not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[(k, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, cs in acc.items():
    ds = [v for (kk, _), v in dur.items() if kk == k]
    print(f"{k}: {len(ds)} dispatches, mean {sum(ds) / len(ds):.3f} ms")
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):.4g}")

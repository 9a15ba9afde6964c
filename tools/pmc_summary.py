#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv per kernel name: sum of each counter, dispatch count, mean duration."""
import csv, sys, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
dur = collections.defaultdict(float)
seen = set()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])[:90]
        rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name].add(r["Dispatch_Id"])
        key = (name, r["Dispatch_Id"])
        if key not in seen and r.get("End_Timestamp") and r.get("Start_Timestamp"):
            seen.add(key)
            dur[name] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
names = sorted(rows, key=lambda n: -dur[n])
counters = sorted({c for n in rows for c in rows[n]})
print("kernel,dispatches,total_us," + ",".join(counters))
for n in names[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print('"%s",%d,%.1f,' % (n, len(cnt[n]), dur[n] / 1e3) + ",".join("%.6g" % rows[n][c] for c in counters))

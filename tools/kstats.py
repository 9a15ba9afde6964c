#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: per-step time per kernel (argv[2] = number of denoising steps)."""
import csv, sys, re
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "ddif" in r["Name"])
print("total ddif kernel time per step: %.3f ms" % (tot / steps / 1e6))
for r in rows[:28]:
    n = re.sub(r"^void ", "", r["Name"]); n = re.sub(r"\(.*", "", n).replace("ddif::", "")
    print("%-58s calls/step %6.1f  avg %8.1f us  per-step %7.3f ms  %5.1f%%" % (n[:58], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
          float(r["TotalDurationNs"]) / steps / 1e6, 100 * float(r["TotalDurationNs"]) / tot))

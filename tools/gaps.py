#!/usr/bin/env python3
"""usage: gaps.py <kernel_trace.csv> [launches_per_step]  -- inter-kernel gap statistics of the sampler loop (the longest run of back-to-back kernels of a trace)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 146
# longest run with every gap < 100 us
best, cur = (0, 0), 0
for i in range(1, len(ks) + 1):
    if i == len(ks) or ks[i][0] - ks[i - 1][1] > 100_000:
        if i - cur > best[1] - best[0]:
            best = (cur, i)
        cur = i
run = ks[best[0]:best[1]]
# drop the first and last step's worth (ramp)
run = run[per:-per] if len(run) > 4 * per else run
gaps = [(run[i + 1][0] - run[i][1]) / 1e3 for i in range(len(run) - 1)]
dur = [(k[1] - k[0]) / 1e3 for k in run]
span = (run[-1][1] - run[0][0]) / 1e3
g = sorted(gaps)
print("run of %d kernels (%.1f steps of %d)  span %.1f us  sum(dur) %.1f (%.1f%%)  sum(gaps) %.1f (%.1f%%)" % (len(run), len(run) / per, per, span, sum(dur), 100 * sum(dur) / span, sum(gaps), 100 * sum(gaps) / span))
print("gap us: min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f  mean %.2f;  negative (overlap): %d" % (g[0], g[len(g) // 10], g[len(g) // 2], g[int(len(g) * 0.9)], g[-1], sum(g) / len(g), sum(1 for x in g if x < 0)))
print("per step: kernels %.3f ms + gaps %.3f ms = %.3f ms" % (sum(dur) / (len(run) / per) / 1e3, sum(gaps) / (len(run) / per) / 1e3, span / (len(run) / per) / 1e3))
# gap following each kernel name
by = collections.defaultdict(list)
for i in range(len(run) - 1):
    by[run[i][2][:90]].append((gaps[i], dur[i]))
print("%-92s %6s %8s %8s" % ("kernel (gap AFTER it)", "n", "dur us", "gap us"))
for k, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1]))[:25]:
    print("%-92s %6d %8.2f %8.2f" % (k, len(v), sum(x[1] for x in v) / len(v), sum(x[0] for x in v) / len(v)))

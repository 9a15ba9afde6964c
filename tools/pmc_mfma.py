#!/usr/bin/env python3
"""MFMA-busy summary from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CU_CYCLES (one pass; SQ has 8
slots).  Per kernel and for the whole run:
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)
i.e. matrix-pipe busy cycles (summed over the SIMDs; = 32 per v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32 --
/opt/skills/guides/MI355X_MICROARCH.md, cycle-constant table) over the busy cycles of the CUs' four SIMDs.  ROCm 7.2 ships
no gfx950 section in derived_counters.xml, so the ratio is formed here from the raw counters (the guide's PMC section).

    tools/pmc_mfma.py <counter_collection.csv> <out.json> [top_n]
"""
import collections
import csv
import json
import re
import sys


def main():
    path, out = sys.argv[1:3]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    val = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            n = re.sub(r"\(.*", "", r["Kernel_Name"])[:120]
            val[n][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[n].add(r["Dispatch_Id"])
            key = (n, r["Dispatch_Id"])
            if key not in seen and r.get("End_Timestamp") and r.get("Start_Timestamp"):
                seen.add(key)
                dur[n] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows, tm, tb = [], 0.0, 0.0
    for n in sorted(val, key=lambda k: -dur[k]):
        m, b = val[n].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), val[n].get("SQ_BUSY_CU_CYCLES", 0.0)
        tm += m
        tb += b
        row = {"kernel": n, "dispatches": len(disp[n]), "total_us": dur[n] / 1e3, "mfma_busy": (m / (4.0 * b)) if b else None}
        row.update({k: v for k, v in val[n].items()})
        rows.append(row)
    res = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES (+ SQ_INSTS_VALU when it fits); mfma_busy = MFMA_BUSY / (4 x BUSY_CU)",
           "whole_run_mfma_busy": (tm / (4.0 * tb)) if tb else None, "kernels": rows[:top]}
    import os
    if os.environ.get("DDIF_BUILD_ID"):
        res["build_id"] = os.environ["DDIF_BUILD_ID"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({"whole_run_mfma_busy": res["whole_run_mfma_busy"], "top": [(r["kernel"][:60], r["mfma_busy"]) for r in rows[:6]]}))


if __name__ == "__main__":
    main()

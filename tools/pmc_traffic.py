#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, TCC slots do not fit both) into the
per-launch HBM traffic of the dominant kernel class, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes:  both counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read -> x2.

    tools/pmc_traffic.py <fetch csv> <write csv>   (raw counter_collection.csv or tools/pmc_summary.py output)
    <out.json> [kernel-name regex]
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    with open(path) as f:
        rd = csv.DictReader(f)
        if "Counter_Name" not in rd.fieldnames:  # already aggregated by tools/pmc_summary.py
            return {re.sub(r"\(.*", "", r["kernel"]): (float(r[counter]), int(r["dispatches"])) for r in rd}
        for r in rd:
            if r["Counter_Name"] != counter:
                continue
            n = re.sub(r"\(.*", "", r["Kernel_Name"])
            tot[n] += float(r["Counter_Value"])
            disp[n].add(r["Dispatch_Id"])
    return {n: (tot[n], len(disp[n])) for n in tot}


def main():
    fetch, write, out = sys.argv[1:4]
    pat = re.compile(sys.argv[4] if len(sys.argv) > 4 else r"conv_mfma_kernel<3,")
    F, W = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    rows, fb, wb, nl = [], 0.0, 0.0, 0
    for n in sorted(F, key=lambda n: -F[n][0]):
        f_kib, nf = F[n]
        w_kib, nw = W.get(n, (0.0, 0))
        row = {"kernel": n[:120], "launches": nf, "fetch_bytes_per_launch": 2.0 * 1024.0 * f_kib / max(nf, 1),
               "write_bytes_per_launch": 1024.0 * w_kib / max(nw, 1)}
        rows.append(row)
        if pat.search(n):
            fb += 2.0 * 1024.0 * f_kib
            wb += 1024.0 * w_kib * (nf / max(nw, 1))
            nl += nf
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes), KiB -> bytes, FETCH_SIZE x2 (gfx950 correction)",
           "class_regex": pat.pattern, "class_launches": nl,
           "class_fetch_bytes_per_launch": fb / max(nl, 1), "class_write_bytes_per_launch": wb / max(nl, 1),
           "class_hbm_bytes_per_launch": (fb + wb) / max(nl, 1), "kernels": rows[:40]}
    import os
    if os.environ.get("DDIF_BUILD_ID"):
        res["build_id"] = os.environ["DDIF_BUILD_ID"]  # sha1 over csrc/ (bench.py build_id()): ties the profile to the build it measured
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))


if __name__ == "__main__":
    main()

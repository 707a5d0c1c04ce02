#!/usr/bin/env python3
"""Summarise rocprofv3 output into small tracked files under profiles/.

    python scripts/pmc_summary.py <tag> <kernel_stats.csv> <fetch counter_collection.csv> <write counter_collection.csv>

FETCH_SIZE / WRITE_SIZE are in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on
gfx950 reports half of the bytes of wide coalesced reads, so the read side is doubled; WRITE_SIZE is used as is
(it matches the known byte count of the K(X*,X) writer kernel 1:1)."""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    tag, stats, fetch, write = sys.argv[1:5]
    out = {"tag": tag, "kernels": {}}
    for r in csv.DictReader(open(stats)):
        k = r["Name"].split("(")[0].replace("void ", "")
        out["kernels"][k] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                             "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
    f, w = per_kernel(fetch), per_kernel(write)
    for k in set(f) | set(w):
        d = out["kernels"].setdefault(k, {})
        if k in f:
            d["FETCH_SIZE_KiB_per_launch"] = f[k][1] / f[k][0]
        if k in w:
            d["WRITE_SIZE_KiB_per_launch"] = w[k][1] / w[k][0]
        d["hbm_bytes_per_launch_corrected"] = (2.0 * d.get("FETCH_SIZE_KiB_per_launch", 0.0)
                                               + d.get("WRITE_SIZE_KiB_per_launch", 0.0)) * 1024.0
    json.dump(out, open("profiles/%s_rocprof_summary.json" % tag, "w"), indent=1, sort_keys=True)
    print("wrote profiles/%s_rocprof_summary.json" % tag)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise rocprofv3 output into small tracked files under profiles/.

    python scripts/pmc_summary.py <tag> <kernel_stats.csv> <fetch counter_collection.csv> <write counter_collection.csv> \
                                  [<SQ / TCC counter_collection.csv> ...]

The optional SQ / TCC passes (one csv per rocprofv3 --pmc run) go to profiles/<tag>_sq_pmc.json: per kernel the
per-launch mean of every counter plus the derived MFMA-pipe busy fraction, L2 hit rate, LDS conflict rate.

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
    if len(sys.argv) > 5:
        pmc = {"tag": tag, "units": "per launch (mean); SQ_*_CYCLES of busy/wave/wait counters are in quad-cycles except "
                                    "SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES (cycles); see MI355X_MICROARCH.md",
               "kernels": {}}
        for path in sys.argv[5:]:
            try:
                rows = list(csv.DictReader(open(path)))
            except Exception:
                continue
            agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
            for r in rows:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                a = agg[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
            for k, cs in agg.items():
                d = pmc["kernels"].setdefault(k, {})
                for c, (n, v) in cs.items():
                    d[c] = v / n
                    d["launches"] = n
        for k, d in pmc["kernels"].items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"]:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the matrix pipes are 256 CUs x 4 SIMDs
                d["kernel_cycles"] = d["GRBM_GUI_ACTIVE"] / 8.0
                d["mfma_pipe_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["kernel_cycles"] * 1024.0)
            if "TCC_HIT_sum" in d and (d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0.0)) > 0:
                d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0.0))
            if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"):
                d["lds_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
        json.dump(pmc, open("profiles/%s_sq_pmc.json" % tag, "w"), indent=1, sort_keys=True)
        print("wrote profiles/%s_sq_pmc.json" % tag)


if __name__ == "__main__":
    main()

#!/bin/bash
# Dev: HBM traffic and SQ counters of k_ei_fused128 (the one-kernel small-N EI pass).   bash scripts/dev/pmc_fused.sh [N M D H]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-128}; M=${2:-200000}; D=${3:-8}; H=${4:-10}
O=$R/gpurun_out/pmc_fused; rm -rf $O; mkdir -p $O
CMD="python $R/scripts/dev/small_n_loop.py $N $M $D $H 12"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o st -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_ei_fused128" --output-format csv -d $O/f -o f -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_ei_fused128" --output-format csv -d $O/w -o w -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "k_ei_fused128" --output-format csv -d $O/a -o a -- $CMD > /dev/null 2>&1
python - <<PY
import csv, glob, collections
N, M, D, H = $N, $M, $D, $H
Dp = 4 if D <= 4 else 8 if D <= 8 else 16 if D <= 16 else (D + 31) // 32 * 32
Mp = (M + 127) // 128 * 128
vals = {}
for tag in "fwa":
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            vals[k] = sum(v) / len(v)
            print("%-32s mean per launch %.4g   (%d launches)" % (k, vals[k], len(v)))
for f in glob.glob("$O/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused" in r["Name"]:
            print(r["Name"][:40], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
            avg = float(r["AverageNs"]) * 1e-9
alg = H * Mp * Dp * 8.0 + H * Mp * 8.0 + H * Mp * 8.0          # scaled candidates + their norms in, EI out
traffic = (2.0 * vals.get("FETCH_SIZE", 0.0) + vals.get("WRITE_SIZE", 0.0)) * 1024.0
print("N=%d M=%d D=%d H=%d: algorithmic bytes per launch (2 cand/ls + |.|^2 in, EI out) %.4g, measured (2 FETCH + WRITE) KiB -> %.4g bytes: x%.2f" % (N, M, D, H, alg, traffic, traffic / alg))
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
    print("matrix pipes busy %.3f of the SIMD-cycles; fp64 MFMA flops executed %.4g (full-square 2 N^2 x evals = %.4g; W is triangular, so about 0.56 of it + the Gram tiles is the expected count)" % (
        vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), vals.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512, 2.0 * N * N * M * H))
PY

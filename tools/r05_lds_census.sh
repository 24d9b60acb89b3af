#!/bin/bash
# LDS census of every kernel of a command: how busy the LDS pipe is and how much of that is bank conflicts.
#   tools/r05_lds_census.sh <name> <python args...>     -> gpurun_out/lds_<name>.txt
# One --pmc pass (SQ + GRBM counters only, no trace domains).  busy = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles), kernel cycles =
# GRBM_GUI_ACTIVE / 8 XCDs; conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
set -u
NAME=$1; shift
OUT=$PWD/gpurun_out/lds_$NAME
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o p -- python "$@" > $OUT/run.log 2>&1
python - "$OUT" <<'PY' | tee $PWD/gpurun_out/lds_$NAME.txt
import csv, glob, collections, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[(row["Kernel_Name"], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
rows = []
for (k, grid), d in agg.items():
    n = max(len(v) for v in d.values())
    m = {c: sum(v) / len(v) for c, v in d.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc <= 0 or m.get("SQ_LDS_IDX_ACTIVE", 0) <= 0:
        continue
    busy = m["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
    conf = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
    wl = m.get("SQ_WAIT_INST_LDS", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)
    wa = m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)
    wi = m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)
    rows.append((n * cyc, k.replace("void pytc::", "")[:78], grid, n, cyc / 2.4e3, busy, conf, wl, wa, wi))
print(f"{'kernel':78s} {'grid':>9s} {'n':>4s} {'us@2.4GHz':>9s} {'LDS busy':>8s} {'conflict':>8s} {'wait LDS':>8s} {'wait any':>8s} {'issue st':>8s}")
for _, k, grid, n, us, busy, conf, wl, wa, wi in sorted(rows, reverse=True):
    print(f"{k:78s} {grid:>9s} {n:4d} {us:9.1f} {busy:8.2f} {conf:8.2f} {wl:8.2f} {wa:8.2f} {wi:8.2f}")
PY
rm -rf $OUT

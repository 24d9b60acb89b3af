#!/bin/bash
# MFMA / VALU / LDS utilisation counters of the bench command's kernels (separate --pmc passes, no trace domains).
set -u
OUT=$PWD/gpurun_out/prof_bench_mfma
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-train"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES --output-format csv -d $OUT/p1 -o b -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/p2 -o b -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/p3 -o b -- $CMD > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
rows = []
for k, cs in agg.items():
    m = lambda c: (sum(cs[c]) / len(cs[c])) if c in cs else float("nan")
    n = len(next(iter(cs.values())))
    rows.append((m("GRBM_GUI_ACTIVE") * n, k, n, cs, m))
rows.sort(key=lambda r: -(r[0] if r[0] == r[0] else 0))
with open("$OUT/summary.txt", "w") as o:
    o.write("kernel | launches | GRBM_GUI_ACTIVE (cycles/launch) | MFMA bf16 MOPS/launch | SQ_VALU_MFMA_BUSY/SQ_BUSY | VALU active/wave cycles | wait_any/wave cycles | LDS conflict/active\n")
    for tot, k, n, cs, m in rows[:16]:
        o.write("%s | %d | %.4g | %.4g | %.3f | %.3f | %.3f | %.3f\n" % (k[:100], n, m("GRBM_GUI_ACTIVE"), m("SQ_INSTS_VALU_MFMA_MOPS_BF16"),
                m("SQ_VALU_MFMA_BUSY_CYCLES") / max(m("SQ_BUSY_CYCLES"), 1), m("SQ_ACTIVE_INST_VALU") / max(m("SQ_WAVE_CYCLES"), 1),
                m("SQ_WAIT_ANY") / max(m("SQ_WAVE_CYCLES"), 1), m("SQ_LDS_BANK_CONFLICT") / max(m("SQ_LDS_IDX_ACTIVE"), 1)))
print(open("$OUT/summary.txt").read()[:2500])
PY

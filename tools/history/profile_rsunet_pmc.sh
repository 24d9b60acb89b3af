#!/bin/bash
# MFMA / LDS counters of the RSUNet kernels (conv3d_tile, conv3d_wgrad_mfma, ...), separate --pmc passes, no trace domains.
set -u
OUT=$PWD/gpurun_out/prof_rsunet_pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$PWD
CMD="python tools/history/rsunet_train_probe.py --steps 3 --gc freeze"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES --output-format csv -d $OUT/p1 -o r -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -o r -- $CMD > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
keep = ("conv3d_tile", "conv3d_wgrad_mfma", "conv3d_thin", "dwconvT3d_generic_vec")
with open("$OUT/summary.txt", "w") as o:
    for k, cs in sorted(agg.items()):
        if not any(t in k for t in keep):
            continue
        line = k[:90] + " launches=%d " % len(next(iter(cs.values()))) + " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items()))
        mf, bz = cs.get("SQ_VALU_MFMA_BUSY_CYCLES"), cs.get("SQ_BUSY_CYCLES")
        if mf and bz:
            line += "  mfma_busy/sq_busy=%.3f" % (sum(mf) / max(sum(bz), 1))
        lc, la = cs.get("SQ_LDS_BANK_CONFLICT"), cs.get("SQ_LDS_IDX_ACTIVE")
        if lc and la:
            line += "  lds_conflict/lds_active=%.3f" % (sum(lc) / max(sum(la), 1))
        o.write(line + "\n")
print(open("$OUT/summary.txt").read()[:3000])
PY
tail -2 $OUT/p1.log | cut -c1-200

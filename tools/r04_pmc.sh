#!/bin/bash
# SQ counters (separate --pmc passes, no trace domains) for kbench targets: tools/r04_pmc.sh <out-name> <kbench targets...>
set -u
NAME=$1; shift
OUT=$PWD/gpurun_out/pmc_$NAME
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc1 -o p -- python tools/r04_kb.py "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc2 -o p -- python tools/r04_kb.py "$@" > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -o p -- python tools/r04_kb.py "$@" > $OUT/pmc3.log 2>&1
python - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
root = sys.argv[1]
for sub in ("pmc1", "pmc2", "pmc3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"][:90], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in sorted(agg.items()):
        n = max(len(v) for v in d.values())
        if n < 3: continue
        print(sub, k, n, {c: f"{sum(v)/len(v):.4e}" for c, v in d.items()})
PY

#!/bin/bash
# SQ counters of the up-block mixer shapes (run on the GPU box); separate --pmc passes, no trace domains.
set -u
OUT=$PWD/gpurun_out/prof_mlp_up
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc1 -o p -- python tools/kbench.py mlp_up > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc2 -o p -- python tools/kbench.py mlp_up > $OUT/pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
for sub in ("pmc1", "pmc2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/prof_mlp_up/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "pw_mlp" in row["Kernel_Name"]:
                agg[(row["Kernel_Name"][:60], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in sorted(agg.items()):
        print(sub, k, {c: f"{sum(v)/len(v):.3e}" for c, v in d.items()})
PY

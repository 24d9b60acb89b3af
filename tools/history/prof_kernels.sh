#!/bin/bash
# rocprofv3 passes for one kbench target (run on the GPU box): kernel trace + PMC counters in SEPARATE runs.
#   tools/history/prof_kernels.sh dwconv   -> gpurun_out/prof_<target>/
set -u
T=${1:-dwconv}
OUT=$PWD/gpurun_out/prof_$T
mkdir -p $OUT
export TMPDIR=/tmp
cd $PWD
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python tools/kbench.py $T > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc1 -o p -- python tools/kbench.py $T > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU -d $OUT/pmc2 -o p -- python tools/kbench.py $T > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc3 -o p -- python tools/kbench.py $T > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum GRBM_GUI_ACTIVE -d $OUT/pmc4 -o p -- python tools/kbench.py $T > $OUT/pmc4.log 2>&1
python tools/history/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

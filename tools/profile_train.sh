#!/bin/bash
# rocprofv3 kernel trace of the training leg only (run on the GPU box). Writes gpurun_out/prof_train/.
set -u
OUT=$PWD/gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python tools/train_probe.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o train -- $CMD > $OUT/trace.log 2>&1
python tools/prof_bench_summary.py $OUT > $OUT/summary.txt 2>&1
head -45 $OUT/summary.txt

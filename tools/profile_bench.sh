#!/bin/bash
# rocprofv3 evidence for the bench command (run on the GPU box): kernel trace + stats, and -- in SEPARATE
# passes -- the HBM byte counters.  Writes gpurun_out/prof_bench/ ; tools/history/prof_summary.py condenses it.
set -u
OUT=$PWD/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-extras"
PMC_CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-train --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $PMC_CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
python tools/prof_bench_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

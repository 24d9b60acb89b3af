#!/bin/bash
# Round-6 evidence (the same passes as rounds 3-5) (run on the GPU box from the repo root): rocprofv3 kernel trace + stats of the bench command with ONE window
# stream (per-kernel averages that agree with bench.py's event timings) and with the default window streams (three since round 4) (overlap timeline), the
# HBM byte counters of the bench in separate --pmc passes, and the same three things for the MedNeXt-S training step.
# Output: gpurun_out/prof_r06/ ; condensed files are copied to profiles/ by hand after review.
set -u
OUT=$PWD/gpurun_out/prof_r06
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-extras"
BENCH1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-train --no-extras"
TRAIN="python tools/train_probe.py"
PYTC_SW_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_1stream/trace -o bench -- $BENCH > $OUT/bench_1stream.log 2>&1
PYTC_SW_STREAMS=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bench_1stream/pmc_fetch -o bench -- $BENCH1 > /dev/null 2>&1
PYTC_SW_STREAMS=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bench_1stream/pmc_write -o bench -- $BENCH1 > /dev/null 2>&1
python tools/prof_bench_summary.py $OUT/bench_1stream > $OUT/bench_1stream_summary.txt 2>&1
python tools/make_hbm_counters_csv.py $OUT/bench_1stream $OUT/bench_hbm_counters.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_streams/trace -o bench -- $BENCH > $OUT/bench_streams.log 2>&1
F=$(find $OUT/bench_streams -name "*kernel_trace.csv" | head -1)
python tools/trace_overlap.py $F 70 > $OUT/overlap_streams.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train/trace -o train -- $TRAIN > $OUT/train.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/train/pmc_fetch -o train -- $TRAIN > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/train/pmc_write -o train -- $TRAIN > /dev/null 2>&1
python tools/prof_bench_summary.py $OUT/train > $OUT/train_summary.txt 2>&1
python tools/make_hbm_counters_csv.py $OUT/train $OUT/train_hbm_counters.csv
for d in bench_1stream bench_streams train; do
  S=$(find $OUT/$d/trace -name "*kernel_stats.csv" | head -1); cp $S $OUT/${d}_kernel_stats.csv
done
# keep the merge-back small: the raw traces stay on the box
rm -rf $OUT/bench_1stream $OUT/bench_streams $OUT/train
grep "^{" $OUT/bench_1stream.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('1 stream ms/8win', d['ms_per_8_windows'])"
grep "^{" $OUT/bench_streams.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default streams ms/8win', d['ms_per_8_windows'])"
head -16 $OUT/overlap_streams.txt
head -14 $OUT/train_summary.txt
ls -la $OUT

set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
python tools/exp_r03_streams.py > gpurun_out/r03/streams.log 2>&1
echo "exp rc=$?"
tail -8 gpurun_out/r03/streams.log
OUT=$PWD/gpurun_out/r03/trace2
rm -rf $OUT; mkdir -p $OUT
PYTC_BENCH_VOLUME=165x336x336 STREAMS=2 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python tools/exp_r03_streams.py > $OUT/log.txt 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/trace_overlap.py $F 80 > gpurun_out/r03/overlap2.txt 2>&1
head -20 gpurun_out/r03/overlap2.txt
rm -rf $OUT

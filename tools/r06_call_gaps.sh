cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/prof_gaps
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o train -- python tools/train_probe.py > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/r06_train_gaps.py $F gpurun_out/r06_train_sequence.txt > gpurun_out/r06_train_gaps.txt 2>&1
grep ms_per_step $OUT/trace.log | cut -c1-160 >> gpurun_out/r06_train_gaps.txt
cat gpurun_out/r06_train_gaps.txt
rm -rf $OUT/trace

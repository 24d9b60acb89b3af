cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in rsunet monai; do
OUT=$PWD/gpurun_out/prof_gaps_$w
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/r06_unet_probe.py $w > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/r06_train_gaps.py $F gpurun_out/r06_${w}_sequence.txt > gpurun_out/r06_${w}_gaps.txt 2>&1
grep train_ms $OUT/trace.log | cut -c1-200 >> gpurun_out/r06_${w}_gaps.txt
head -12 gpurun_out/r06_${w}_gaps.txt
rm -rf $OUT/trace
done

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_train.py tests/test_gpu_mednext_2d.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -8
for i in 1 2; do timeout 300 python tools/train_probe.py 2>&1 | grep ms_per_step | cut -c1-120; done
bash tools/r06_call_gaps.sh > /dev/null 2>&1; head -8 gpurun_out/r06_train_gaps.txt) > gpurun_out/r06_tail.log 2>&1
cat gpurun_out/r06_tail.log

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_training.py -x -q -k "rowmajor" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
for v in 0 16384 100000 0 100000; do
echo "== PYTC_TRAIN_GEMM_MAX_ROWS=$v"
PYTC_TRAIN_GEMM_MAX_ROWS=$v timeout 300 python tools/train_probe.py 2>&1 | grep ms_per_step | cut -c1-110
done) 2>&1 | grep -v amdgpu > gpurun_out/r06_gemm_train.log
cat gpurun_out/r06_gemm_train.log

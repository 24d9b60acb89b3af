cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_training.py tests/test_gpu_baseline_sizes.py tests/test_gpu_odd_shapes.py -q 2>&1 | tail -8
for v in 0 1 0 1; do
echo "== train step, PYTC_UP_NORM_STATS_FROM_WGRAD=$v"
PYTC_UP_NORM_STATS_FROM_WGRAD=$v timeout 600 python tools/train_probe.py 2>&1 | grep -v amdgpu | grep -E "ms_per_step" | cut -c1-120
done

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py -q -k "dw_wgrad or depthwise_backward" 2>&1 | tail -3
for v in 0 1024 0 1024; do
echo "== train step, dw_wgrad_small_wgs=$v"
PYTC_TUNING=dw_wgrad_small_wgs=$v PROBE_TOP=100 timeout 600 python tools/train_probe.py --ops 2>&1 | grep -v amdgpu | grep -E "ms_per_step|dw_wgrad\[" | cut -c1-130
done

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_training.py -q -k "fused_weight_gradient or rebuilt or c2_training" 2>&1 | tail -3
for v in 0 1 0 1; do
echo "== train step, wgrad_dgrad_wide=$v"
PYTC_TUNING=wgrad_dgrad_wide=$v PROBE_TOP=100 timeout 600 python tools/train_probe.py --ops 2>&1 | grep -v amdgpu | grep -E "ms_per_step|128->32|32->128" | cut -c1-130
done

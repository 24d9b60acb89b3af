cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
echo "== train step, PYTC_RC_FWD_F16_GELU=$v"
PYTC_RC_FWD_F16_GELU=$v PROBE_TOP=6 timeout 600 python tools/train_probe.py --ops 2>&1 | grep -v amdgpu | grep -E "ms_per_step|nostore|total kernel" | cut -c1-160
done
PYTC_RC_FWD_F16_GELU=1 timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_baseline_sizes.py -q -k "c2_training or baseline or rebuilt" 2>&1 | tail -5

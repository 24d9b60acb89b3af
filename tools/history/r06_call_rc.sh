# round 6: mixer backward with the hidden pre-activation rebuilt -- tests, then the training step with the switch off / on
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_training.py tests/test_gpu_baseline_sizes.py tests/test_gpu_fused_train.py -q 2>&1 | tail -15
echo "== train step, PYTC_MIXER_BWD_RC=0"
PYTC_MIXER_BWD_RC=0 timeout 600 python tools/train_probe.py 2>&1 | grep -v amdgpu | tail -2 | cut -c1-120
echo "== train step, PYTC_MIXER_BWD_RC=1"
PYTC_MIXER_BWD_RC=1 PROBE_TOP=12 timeout 600 python tools/train_probe.py --ops 2>&1 | grep -v amdgpu | tail -14 | cut -c1-160

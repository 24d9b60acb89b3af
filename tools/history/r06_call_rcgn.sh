cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
echo "== train step, mixer_bwd_rc_gn=$v"
PYTC_TUNING=mixer_bwd_rc_gn=$v PROBE_TOP=100 timeout 600 python tools/train_probe.py --ops 2>&1 | grep -v amdgpu | grep -E "ms_per_step|mixer_bwd_rc|pw_wgrad_gn\[32" | cut -c1-130
done

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1; do
echo "== mixer_bwd_rc_probe=$v"
PYTC_TUNING=mixer_bwd_rc_probe=$v PROBE_TOP=100 timeout 600 python tools/train_probe.py --ops 2>&1 | grep -v amdgpu | grep -E "mixer_bwd_rc" | cut -c1-130
done

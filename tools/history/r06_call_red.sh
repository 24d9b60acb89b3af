cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 1 2 1 2; do
echo "== mixer_bwd_rc_slot_div=$v"
PYTC_TUNING=mixer_bwd_rc_slot_div=$v PROBE_TOP=100 timeout 600 python tools/train_probe.py --ops 2>&1 | grep -v amdgpu | grep -E "ms_per_step|reduce_slots_multi|mixer_bwd_rc|total kernel" | cut -c1-130
done

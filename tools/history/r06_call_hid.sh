cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 1000000 512 256 1000000 512; do
echo "== PYTC_FUSED_MIXER_MAX_HID=$v"
PYTC_FUSED_MIXER_MAX_HID=$v timeout 300 python tools/train_probe.py 2>&1 | grep ms_per_step | cut -c1-110
done

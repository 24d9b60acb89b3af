# A/B of two library builds on the headline bench: lib A = in-tree build, lib B = pytorch_connectomics_amd/lib/libpytc_hip_B.so
set -u
mkdir -p gpurun_out/r03
L=pytorch_connectomics_amd/lib
for tag in A B A B; do
  if [ $tag = B ]; then cp $L/libpytc_hip.so /tmp/keepA.so; cp $L/libpytc_hip_B.so $L/libpytc_hip.so; fi
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --train-steps 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', 'ms_per_8_windows', round(d.get('ms_per_8_windows', 0), 3), 'train_ms', d.get('train', {}).get('ms_per_step'))"
  if [ $tag = B ]; then cp /tmp/keepA.so $L/libpytc_hip.so; fi
done

#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_call11
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mednext.py tests/test_gpu_monai_unet.py tests/test_gpu_baseline_sizes.py -x -q -k "merged or monai or c4 or c5 or strided or unet" > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
for m in 0 1; do echo "== PYTC_MERGE_HEADS=$m"; PYTC_MERGE_HEADS=$m timeout 300 python tools/r05_l_forward.py 3 2 2>&1 | grep -v amdgpu | head -14; done > $OUT/l_forward.txt 2>&1; cat $OUT/l_forward.txt | cut -c1-200
for k in 0 1; do echo "== conv3d_wgrad_c1_line=$k"; PYTC_TUNING="conv3d_wgrad_c1_line=$k" timeout 300 python tools/r03_unet_legs.py --roofline 2>&1 | grep -v amdgpu | head -3; done > $OUT/unet.txt 2>&1; cat $OUT/unet.txt | cut -c1-600

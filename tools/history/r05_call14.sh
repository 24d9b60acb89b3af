#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_call14
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_fused_train.py -x -q > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
for f in 0 1; do echo "== PYTC_FUSED_WGRAD_DGRAD=$f"; PYTC_FUSED_WGRAD_DGRAD=$f timeout 300 python tools/train_probe.py --ops 2>&1 | grep -v amdgpu | head -16 | cut -c1-230; done > $OUT/train_ab.txt 2>&1; cat $OUT/train_ab.txt

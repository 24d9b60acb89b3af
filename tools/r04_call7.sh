#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 200 python tools/r04_resample_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04_resample2.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_mednext.py tests/test_gpu_training.py -q -x 2>&1 | tail -3
timeout 900 bash tools/r04_ab.sh "PYTC_TUNING=dwconv_s2_march=0" "PYTC_TUNING=dwconv_s2_march=1" 2>&1 | tee gpurun_out/r04_s2_ab.txt

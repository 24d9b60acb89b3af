#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 200 python tools/r04_resample_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04_resample.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mednext.py -q -x 2>&1 | tail -3
timeout 900 bash tools/r04_ab.sh "PYTC_TUNING=dwconvT_tile=0" "PYTC_TUNING=dwconvT_tile=1" 2>&1 | tee gpurun_out/r04_dwT_ab.txt

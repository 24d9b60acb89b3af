#!/bin/bash
# round 4, LDS-resident mixer: parity tests, micro-benchmark, whole-volume A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "lds_resident or deep_level_gemm" 2>&1 | tail -4
timeout 300 python -m pytest tests/test_gpu_mednext.py -q -x 2>&1 | tail -3
timeout 300 python tools/r04_lds_mixer.py > gpurun_out/r04_lds_mixer.txt 2>&1; tail -22 gpurun_out/r04_lds_mixer.txt
timeout 900 bash tools/r04_ab.sh "PYTC_LDS_MIXER_ROWS=0" "PYTC_LDS_MIXER_ROWS=131072" "PYTC_LDS_MIXER_ROWS=131072 PYTC_SW_STREAMS=1" "PYTC_LDS_MIXER_ROWS=0 PYTC_SW_STREAMS=1" 2>&1 | tee gpurun_out/r04_lds_ab.txt

#!/bin/bash
# Round 5, call 2: the conflict-free LDS layout of the matrix-core depthwise conv: parity tests, timings, LDS counters, phases.
set -u
OUT=$PWD/gpurun_out/r05_call2
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "dwconv or depthwise or mfma or march" > $OUT/pytest_dw.txt 2>&1; tail -5 $OUT/pytest_dw.txt
timeout 600 python -m pytest tests/test_gpu_training.py -x -q -k "balancing" > $OUT/pytest_bal.txt 2>&1; tail -5 $OUT/pytest_bal.txt
timeout 300 python tools/r04_dwmfma.py > $OUT/dwmfma.txt 2>&1; tail -12 $OUT/dwmfma.txt
timeout 200 python tools/r05_additivity.py phases > $OUT/phases.txt 2>&1; cat $OUT/phases.txt
timeout 400 bash tools/r04_pmc.sh r05_dw dw0 dw1 > $OUT/pmc_dw.txt 2>&1
grep "dwconv3d_k3_mfma" gpurun_out/pmc_r05_dw/summary.txt > $OUT/pmc_dw_summary.txt; cat $OUT/pmc_dw_summary.txt | cut -c1-700
rm -rf gpurun_out/pmc_r05_dw

#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_call4
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dwmix.py -x -q > $OUT/pytest_dwmix.txt 2>&1; tail -15 $OUT/pytest_dwmix.txt
timeout 300 python tools/r05_dwmix_bench.py > $OUT/dwmix_bench.txt 2>&1; cat $OUT/dwmix_bench.txt

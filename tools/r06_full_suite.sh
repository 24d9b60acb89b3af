#!/bin/bash
# the whole GPU suite + smoke on the current tree (the record the round-end driver run is compared with)
set -u
OUT=$PWD/gpurun_out/r06_suite
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt

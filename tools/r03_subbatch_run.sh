#!/bin/bash
# One gpurun call: level-0 sub-batch A/B (tools/exp_r03_subbatch.py), the MedNeXt GPU tests (incl. the bit-identity test of the
# sub-batched schedule), then the BASELINE-size and window tests and a short bench line under the best configuration.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 240 python tools/exp_r03_subbatch.py > gpurun_out/r03_subbatch.log 2>&1
echo "exp rc=$?" >> gpurun_out/r03_subbatch.log
timeout 200 python -m pytest tests/test_gpu_mednext.py -x -q > gpurun_out/r03_pytest_mednext.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_pytest_mednext.log
BEST=$(python -c "import json; print(json.load(open('gpurun_out/r03_subbatch_best.json'))['l0_subbatch'])" 2>/dev/null || echo 0)
echo "best l0_subbatch=$BEST" >> gpurun_out/r03_subbatch.log
PYTC_L0_SUBBATCH=$BEST timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-extras > gpurun_out/r03_bench_best.log 2>&1
PYTC_L0_SUBBATCH=$BEST timeout 300 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_window.py -x -q > gpurun_out/r03_pytest_best.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_pytest_best.log
tail -3 gpurun_out/r03_subbatch.log gpurun_out/r03_pytest_mednext.log gpurun_out/r03_bench_best.log gpurun_out/r03_pytest_best.log

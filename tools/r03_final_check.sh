#!/bin/bash
# Last GPU call of the round (2.8 GPU-minutes left): the bench line through the new headline / watchdog code, the lazy TTA + mask
# path and the TTA predictor on the real kernels, smoke().
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 70 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --train-steps 2 > gpurun_out/r03_final_bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/r03_final_bench.log
timeout 70 python -m pytest tests/test_gpu_zz_lazy_tta.py tests/test_gpu_tta.py -x -q > gpurun_out/r03_final_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_final_pytest.log
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_final_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r03_final_smoke.log
tail -n 3 gpurun_out/r03_final_pytest.log; tail -n 2 gpurun_out/r03_final_smoke.log; tail -c 300 gpurun_out/r03_final_bench.log

#!/bin/bash
# Second short GPU check of the session: the tests that cross the modules re-structured / extended on the CPU side (registry, build,
# manager, stage, model_outputs, mask application, lazy TTA) with the real kernels.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_tta.py tests/test_gpu_zz_lazy_tta.py tests/test_gpu_lazy_chunked.py tests/test_gpu_main.py -x -q --durations=5 > gpurun_out/r03_final_pytest2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_final_pytest2.log
tail -n 12 gpurun_out/r03_final_pytest2.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_mednext.py tests/test_gpu_monai_unet.py::test_monai_unet_training_step_matches_oracle_autograd tests/test_gpu_baseline_sizes.py::test_c2_bench_path_mednext_s_112_engine_vs_oracle -q -x ) > gpurun_out/r02_pytest_d.log 2>&1
tail -25 gpurun_out/r02_pytest_d.log
( timeout 300 python tools/exp_r02.py upfuse ) > gpurun_out/r02_exp_up.log 2>&1
cat gpurun_out/r02_exp_up.log

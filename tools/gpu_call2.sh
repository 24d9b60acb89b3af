cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_monai_unet.py tests/test_gpu_baseline_sizes.py::test_c4_chunked_real_geometry_is_exact tests/test_gpu_baseline_sizes.py::test_c4_chunked_mednext_l_96_crop_vs_oracle -q ) > gpurun_out/r02_pytest_b.log 2>&1
tail -30 gpurun_out/r02_pytest_b.log
( timeout 300 python tools/exp_r02.py nsweep lut ) > gpurun_out/r02_exp.log 2>&1
cat gpurun_out/r02_exp.log
( time timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_baseline_sizes.py --deselect tests/test_gpu_monai_unet.py ) > gpurun_out/r02_pytest_c.log 2>&1
tail -5 gpurun_out/r02_pytest_c.log

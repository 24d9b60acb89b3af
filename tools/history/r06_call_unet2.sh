timeout 600 python -m pytest tests/test_gpu_monai_unet.py tests/test_gpu_kernels.py tests/test_gpu_rsunet.py -x -q -k "strided or convT or monai or transposed or resample or conv3d or rsunet" 2>&1 | tail -4
python tools/r06_unet_labels.py monai 14 2>&1 | grep -v amdgpu

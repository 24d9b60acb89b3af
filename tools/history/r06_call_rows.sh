cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_monai_unet.py tests/test_gpu_kernels.py tests/test_gpu_rsunet.py tests/test_gpu_rsunet_training.py -x -q -k "strided or convT or monai or transposed or resample or conv3d or rsunet or conv" 2>&1 | tail -4
python tools/r06_conv_tile_probe.py --rsunet 2>&1 | grep -v amdgpu
python tools/r06_conv_tile_probe.py 2>&1 | grep -v amdgpu
python tools/r06_unet_labels.py monai 8 2>&1 | grep -v amdgpu | head -4
python tools/r06_unet_labels.py rsunet 12 2>&1 | grep -v amdgpu | head -16

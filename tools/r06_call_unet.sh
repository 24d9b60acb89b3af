timeout 600 python -m pytest tests/test_gpu_monai_unet.py tests/test_gpu_kernels.py -x -q -k "strided or convT or monai or transposed" 2>&1 | tail -4
for pm in 0 1; do echo "== convT_phase_major=$pm"; PYTC_TUNING=convT_phase_major=$pm python tools/r03_unet_legs.py --roofline 2>&1 | grep -v amdgpu | head -3 | cut -c1-900; done

#!/bin/bash
# round 6: the stride-2 transposed gathers of the MONAI-style U-Net -- gather form in raster / phase-major row order, and the LDS-tiled phase form
timeout 600 python -m pytest tests/test_gpu_monai_unet.py tests/test_gpu_kernels.py -x -q -k "strided or convT or monai or transposed or resample" 2>&1 | tail -4
for cfg in "convT_phase_tile=0,convT_phase_major=0" "convT_phase_tile=0,convT_phase_major=1" "convT_phase_tile=1"; do
  echo "== $cfg"; PYTC_TUNING=$cfg python tools/r03_unet_legs.py --roofline 2>&1 | grep -v amdgpu | head -2 | cut -c1-1100
done

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for pd in 1 2 1 2; do
echo "== MedNeXt-L forward, pw_gemm_pd=$pd"
PYTC_TUNING=pw_gemm_pd=$pd timeout 600 python tools/r05_l_forward.py 6 2 2>&1 | grep -v amdgpu | grep -E "ms per forward|pw_gemm" | cut -c1-150
done
echo "== MedNeXt-S bench, pd 1 / 2"
for pd in 1 2 1 2; do
PYTC_TUNING=pw_gemm_pd=$pd python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-train --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('pd', $pd, 'ms/8win', round(d['ms_per_8_windows'], 3))"
done
PYTC_TUNING=pw_gemm_pd=2 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -3

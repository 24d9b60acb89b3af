cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/probes/bin/gelu_probe > gpurun_out/r02_gelu_probe.txt 2>&1
cat gpurun_out/r02_gelu_probe.txt
( time timeout 600 python -m pytest tests/test_gpu_main.py tests/test_gpu_lazy_chunked.py -q -x ) > gpurun_out/r02_pytest_e.log 2>&1
tail -15 gpurun_out/r02_pytest_e.log

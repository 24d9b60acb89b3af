cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02_pytest_gpu.log 2>&1
tail -5 gpurun_out/r02_pytest_gpu.log
( time timeout 600 python bench.py --steps 5 --warmup 1 ) > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err
tail -c 3000 gpurun_out/r02_bench_a.json; tail -5 gpurun_out/r02_bench_a.err
OUT=$PWD/gpurun_out/prof_bench_a; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-extras > $OUT/trace.log 2>&1
ls $OUT/trace/* | head; 

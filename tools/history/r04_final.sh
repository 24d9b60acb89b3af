#!/bin/bash
# round 4, final evidence run: the default bench line, then the rocprofv3 passes of tools/profile_r04.sh
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r04_bench_final.log 2>&1
grep "^{" gpurun_out/r04_bench_final.log | tail -1 > gpurun_out/r04_bench_line.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_line.json'))
print('value', d['value'], 'ms/vol', d['ms_per_step'], 'ms/8win', d['ms_per_8_windows'], 'roof', d['roofline']['kernel'], d['roofline']['frac'], 'whole', d['roofline']['whole_step']['frac'], 'train', d['train']['ms_per_step'])"
timeout 1200 bash tools/profile_r04.sh 2>&1 | tail -40

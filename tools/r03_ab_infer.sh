#!/bin/bash
# A/B of the inference engine (whole-volume bench, no extras): PYTC_MIXER_TWO_GEMM_ROWS in {0, 16384, 32768}, 2 and 1 window streams
for rows in 0 16384 32768 0 16384; do
  for st in 2 1; do
    PYTC_MIXER_TWO_GEMM_ROWS=$rows PYTC_SW_STREAMS=$st python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-extras 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two_gemm_rows=$rows streams=$st ms/8win %.3f' % d['ms_per_8_windows'])"
  done
done

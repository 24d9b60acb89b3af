#!/bin/bash
# round 6: the headline step against the level-0 sub-batch (PYTC_L0_SUBBATCH: depth-first sample slices at full resolution, so that a slice's
# depthwise output could be re-read from the 256 MiB Infinity Cache) and the number of window streams; one box
for st in 3 1 2; do
  for sb in 0 2 1 4; do
    PYTC_SW_STREAMS=$st PYTC_L0_SUBBATCH=$sb python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-train --no-extras 2>/dev/null | grep "^{" | tail -1 | \
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print('streams', $st, 'l0_subbatch', $sb, 'ms/8win', round(d['ms_per_8_windows'], 3))"
  done
done

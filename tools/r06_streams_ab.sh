#!/bin/bash
# round 6: the headline step against the number of window streams (PYTC_SW_STREAMS), two passes in alternating order, one box
for pass in 1 2; do
  for st in 3 4 2 6; do
    PYTC_SW_STREAMS=$st python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-train --no-extras 2>/dev/null | grep "^{" | tail -1 | \
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print('streams', $st, 'ms/8win', round(d['ms_per_8_windows'], 3))"
  done
done

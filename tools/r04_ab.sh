#!/bin/bash
# A/B of the headline step under environment switches, same box, alternating order:  tools/r04_ab.sh "A=1" "A=0" ...
# prints ms_per_8_windows of `bench.py --steps 4 --warmup 2` (inference only) for every setting, two rounds
for round in 1 2; do
  for setting in "$@"; do
    ms=$(env $setting python bench.py --steps 4 --warmup 2 --no-extras --no-train --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_8_windows'],3), d['config']['window_pipeline_streams'])")
    echo "round $round  $setting  ms_per_8_windows $ms"
  done
done

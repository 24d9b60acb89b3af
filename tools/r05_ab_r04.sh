#!/bin/bash
# Same-box A/B of the whole step: this tree against the round-4 tree (a git worktree of 1e44846 built under _r04/, git-ignored).
#   tools/r05_ab_r04.sh [extra bench flags]
B="--steps 4 --warmup 2 --no-extras --no-train --no-cpu-baseline --no-roofline"
for round in 1 2; do
  for tree in _r04 .; do
    for streams in 3 1; do
      ms=$(cd $tree && PYTC_SW_STREAMS=$streams python bench.py $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_8_windows'],3))")
      echo "round $round  tree $tree  streams $streams  ms_per_8_windows $ms"
    done
  done
done

#!/bin/bash
# Same-box A/B of the whole step: this tree against a worktree of the last commit (`git worktree add _prev HEAD`, built; git-ignored).
#   tools/r05_ab_prev.sh [extra bench flags]      also prints MedNeXt-L's forward in both trees
B="--steps 4 --warmup 2 --no-extras --no-train --no-cpu-baseline --no-roofline"
for round in 1 2; do
  for tree in _prev .; do
    ms=$(cd $tree && python bench.py $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_8_windows'],3))")
    echo "round $round  tree $tree  ms_per_8_windows $ms"
  done
done
for tree in _prev . _prev .; do
  (cd $tree && python tools/r05_l_forward.py 2>&1 | grep "ms per forward" | head -1 | cut -c1-110 | sed "s/^/tree $tree  /")
done

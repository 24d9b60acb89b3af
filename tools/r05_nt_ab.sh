#!/bin/bash
# nontemporal stream policy A/B: isolated level-0 kernels (tools/r04_kb.py) and the whole step, same box, alternating
for k in 0 1; do
  echo "== stream_nt=$k"
  PYTC_TUNING="stream_nt=$k" python tools/r04_kb.py mix0 up0 mix1 up1 2>&1 | grep -v amdgpu.ids
done
bash tools/r04_ab.sh "PYTC_TUNING=stream_nt=0 PYTC_FUSE_BLOCK=0" "PYTC_TUNING=stream_nt=1 PYTC_FUSE_BLOCK=0" "PYTC_FUSE_BLOCK=0"

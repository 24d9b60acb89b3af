#!/bin/bash
# round 4, call 1: the Toeplitz depthwise probe (VERDICT r03 item 1a gate) + isolated mixer timings as this round's baseline
mkdir -p gpurun_out
( timeout 120 tools/probes/bin/toeplitz_dwconv_probe check; timeout 180 tools/probes/bin/toeplitz_dwconv_probe time 8 ) > gpurun_out/r04_toeplitz_probe.txt 2>&1
timeout 300 python tools/kbench.py mlp > gpurun_out/r04_kbench_mlp.txt 2>&1
tail -30 gpurun_out/r04_toeplitz_probe.txt; tail -40 gpurun_out/r04_kbench_mlp.txt

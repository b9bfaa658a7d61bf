#!/bin/bash
# does the issue-rate microbenchmark see the idle-clock effect?  the same ops with 1 and with 40 warm-up launches
cd "$(dirname "$0")/../../.."
mkdir -p gpurun_out/peakw
for W in 1 40 1 40; do for k in k_add k_fma k_cndmask_vcc k_min k_boxmix; do ./build/issue_peak -k $k -W $W | grep -v "^# [nmc]\|^op "; done; done | tee gpurun_out/peakw/peak_warm.txt

#!/bin/bash
# phase breakdown of the split convolution's blocks (measurement build EMO_S_TIMING=1), bf16x3 and f16x2, 16 frames
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing.so timeout 300 python tools/conv_phase_timing.py 16 > gpurun_out/r4_conv_phase_timing.jsonl 2> gpurun_out/r4_conv_phase_timing.err
tail -3 gpurun_out/r4_conv_phase_timing.err
cut -c1-600 gpurun_out/r4_conv_phase_timing.jsonl

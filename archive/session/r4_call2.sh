#!/bin/bash
# row-layout epilogue of the split kernel (conv_epilogue_rows): full GPU suite, phase breakdown, layer microbenchmark
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r4_c2_tests.txt; tail -4 gpurun_out/r4_c2_tests.txt
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing.so timeout 300 python tools/conv_phase_timing.py 16 > gpurun_out/r4_c2_phase.jsonl 2> gpurun_out/r4_c2_phase.err
tail -2 gpurun_out/r4_c2_phase.err
timeout 300 python tools/bench_conv.py 16 --quick --bf16x3-only --f16x2 > gpurun_out/r4_c2_convbench.jsonl 2> gpurun_out/r4_c2_convbench.err
cut -c1-400 gpurun_out/r4_c2_convbench.jsonl

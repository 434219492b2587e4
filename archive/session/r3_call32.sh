#!/bin/bash
# what a one-wave-per-SIMD bf16 MFMA stream reaches with the pieces of the bf16x3 K loop added one at a time
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 120 tools/microbench/mfma_stream > gpurun_out/r3_mfma_stream.jsonl 2>&1
cat gpurun_out/r3_mfma_stream.jsonl

#!/bin/bash
# round 5, call 18: the final code -- the whole GPU suite + smoke, then the evidence set (bench line with extras, bf16 split,
# 2-rank functional run, conv microbenchmark, driver breakdowns, kernel trace + PMC passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/gpu_validate.sh r5 tests
bash tools/gpu_validate.sh r5 bench-short

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 300 python tools/profile_source_pass.py 512 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_source_pass_launches.jsonl
cat gpurun_out/r3_source_pass_launches.jsonl

#!/bin/bash
# last check of the tree as committed: conv parity + smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_conv_bf16x3_gpu.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
echo default; timeout 120 python tools/dbg_conv.py 2>&1 | grep -v amdgpu
echo nonpersistent; EMO_CONV_BF16X3_PERSISTENT=0 timeout 120 python tools/dbg_conv.py 2>&1 | grep -v amdgpu

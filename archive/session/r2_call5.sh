#!/bin/bash
# round-2 GPU call 5: tests incl. the 64x512 conv tile; bench default vs EMO_CONV_CFG_E=1; conv microbench with cfg 4; 2 ranks on 1 GPU
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2c5}
mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${T}_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED\|error" gpurun_out/${T}_pytest_full.log > gpurun_out/${T}_pytest.log
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
EMO_CONV_CFG_E=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_cfgE.json 2>> gpurun_out/${T}_bench.err
timeout 300 python tools/bench_conv.py 16 --quick 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv.jsonl
EMO_DIST_BACKEND=gloo EMO_FORCE_DEVICE=0 timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 4 --no-cpu-baseline > gpurun_out/${T}_bench_gpus2.json 2> gpurun_out/${T}_bench_gpus2.err
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_smoke.log
tail -4 gpurun_out/${T}_pytest.log; cut -c1-200 gpurun_out/${T}_bench.json; cut -c1-200 gpurun_out/${T}_bench_cfgE.json; tail -1 gpurun_out/${T}_smoke.log

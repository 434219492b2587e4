#!/bin/bash
# round 3, call 6: after retiring the losing sampler variants (bricks default for the uv call): full parity suite, sampler
# microbench incl. the NCDHW seam routes, bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 1700 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r3c6_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED\|error" gpurun_out/r3c6_pytest_full.log > gpurun_out/r3c6_pytest.log
tail -4 gpurun_out/r3c6_pytest.log
timeout 600 python tools/bench_sampler.py 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c6_sampler.jsonl
timeout 900 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r3c6_bench.json 2> gpurun_out/r3c6_bench.err
cut -c1-300 gpurun_out/r3c6_bench.json

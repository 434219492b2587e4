#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r1_call3_pytest_full.log; grep -a "PARITY\|passed\|failed\|Error" gpurun_out/r1_call3_pytest_full.log > gpurun_out/r1_call3_pytest.log
timeout 300 python tools/bench_sampler.py 8 64 > gpurun_out/r1_call3_sampler.jsonl 2>&1
timeout 600 python tools/bench_driver.py 512 1 4 8 16 > gpurun_out/r1_call3_driver512.jsonl 2>&1
timeout 300 python tools/bench_driver.py 256 1 8 32 > gpurun_out/r1_call3_driver256.jsonl 2>&1
timeout 300 python tools/bench_conv.py 4 > gpurun_out/r1_call3_conv.jsonl 2>&1
tail -12 gpurun_out/r1_call3_pytest.log; cat gpurun_out/r1_call3_driver512.jsonl

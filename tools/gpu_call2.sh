#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -60 > gpurun_out/r1_call2_pytest.log
timeout 300 python tools/bench_sampler.py 8 64 > gpurun_out/r1_call2_sampler.jsonl 2>&1
timeout 600 python tools/bench_conv.py 4 > gpurun_out/r1_call2_conv.jsonl 2>&1
tail -8 gpurun_out/r1_call2_pytest.log

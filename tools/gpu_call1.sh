#!/bin/bash
# first GPU contact: sampler parity + variant microbench + stock torch-ROCm probe
mkdir -p gpurun_out
{ lscpu | head -25; nproc; free -g | head -2; ls /root/reference 2>&1 | head -3; rocm-smi --showproductname 2>&1 | head -12; } > gpurun_out/host.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r1_call1_pytest.log
timeout 600 python tools/bench_sampler.py 1 8 64 > gpurun_out/r1_call1_sampler.jsonl 2>&1
timeout 600 python tools/probe_torch_gpu.py 4 > gpurun_out/r1_call1_torch_probe.jsonl 2>&1
tail -5 gpurun_out/r1_call1_pytest.log

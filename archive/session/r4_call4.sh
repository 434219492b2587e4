#!/bin/bash
# new tests (2-rank product path, verify_checkpoint, graphs by default, sampler tuning-word check), per-wave barrier accounting of
# the split kernel's K loop (EMO_S_TIMING=2), bench step in bf16x3 and guarded f16x2
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_two_ranks_gpu.py tests/test_verify_checkpoint_gpu.py tests/test_infer_gpu.py tests/test_grid_sample_gpu.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r4_c4_tests.txt; tail -5 gpurun_out/r4_c4_tests.txt
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing2.so timeout 200 python tools/conv_phase_timing.py 16 > gpurun_out/r4_c4_phase_waves.jsonl 2> gpurun_out/r4_c4_phase.err
tail -2 gpurun_out/r4_c4_phase.err
for p in bf16x3 f16x2; do
  EMO_CONV_PRECISION=$p timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c4_bench_$p.err | tee gpurun_out/r4_c4_bench_$p.json | cut -c1-420
done

#!/bin/bash
# per-wave barrier accounting of the split kernel's K loop (EMO_S_TIMING=2), bench step in guarded f16x2
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing2.so timeout 200 python tools/conv_phase_timing.py 16 > gpurun_out/r4_c5_phase_waves.jsonl 2> gpurun_out/r4_c5_phase.err
tail -2 gpurun_out/r4_c5_phase.err
for p in f16x2; do
  EMO_CONV_PRECISION=$p timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c5_bench_$p.err | tee gpurun_out/r4_c5_bench_$p.json | cut -c1-300
  tail -2 gpurun_out/r4_c5_bench_$p.err
done

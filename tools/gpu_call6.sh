#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r1_call6_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED" gpurun_out/r1_call6_pytest_full.log > gpurun_out/r1_call6_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r1_call6_smoke.log 2>&1
for v in "" endstore kc2 kc8; do
  if [ -z "$v" ]; then unset EMO_HIP_LIB; tag=default; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so; tag=$v; fi
  timeout 300 python tools/bench_conv.py 4 --quick > gpurun_out/r1_call6_conv_$tag.jsonl 2>&1
  timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r1_call6_bench_$tag.json 2>&1
done
unset EMO_HIP_LIB
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1_call6_prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1_call6_prof_write.log 2>&1
cd $R
for p in fetch write; do python tools/summarize_rocprof.py pmc gpurun_out/prof_$p gpurun_out/r1_pmc16_$p.json; done
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
tail -6 gpurun_out/r1_call6_pytest.log; tail -2 gpurun_out/r1_call6_smoke.log; for t in default endstore kc2 kc8; do echo $t; tail -1 gpurun_out/r1_call6_bench_$t.json | cut -c1-220; done

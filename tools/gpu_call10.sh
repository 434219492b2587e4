#!/bin/bash
# parity (all GPU tests), smoke, bench line, rocprofv3 kernel stats + PMC passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r1_call10_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED" gpurun_out/r1_call10_pytest_full.log > gpurun_out/r1_call10_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r1_call10_smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/r1_call10_bench.json 2> gpurun_out/r1_call10_bench.err
timeout 300 python bench.py --image-size 256 --batch 32 --no-cpu-baseline > gpurun_out/r1_call10_bench256.json 2>> gpurun_out/r1_call10_bench.err
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1_call10_prof_kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1_call10_prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1_call10_prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_mfma -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1_call10_prof_mfma.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof_kt gpurun_out/r1_kernel_stats.csv
for p in fetch write mfma; do python tools/summarize_rocprof.py pmc gpurun_out/prof_$p gpurun_out/r1_pmc_$p.json; done
find gpurun_out/prof_kt gpurun_out/prof_fetch -type f | head -20 > gpurun_out/r1_call10_prof_files.txt
for f in $(find gpurun_out/prof_fetch -name "*.csv" | head -3); do echo "== $f"; head -5 $f; done >> gpurun_out/r1_call10_prof_files.txt
# keep the merge small: drop raw traces
du -sh gpurun_out/prof_* >> gpurun_out/r1_call10_prof_files.txt
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma
cat gpurun_out/r1_call10_pytest.log | tail -15; cat gpurun_out/r1_call10_smoke.log | tail -3; cat gpurun_out/r1_call10_bench.json

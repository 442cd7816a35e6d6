#!/bin/bash
# round-2 GPU call B: failing scale tests with output, fp32 mode tests, c3 rocprofv3 kernel stats
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -rP -k "llama_layers or temporal_pooling" 2>&1 | grep -v "^$" | tail -120 > gpurun_out/b_scale.log
grep -n "rel-L2\|passed\|failed\|Error\|assert" gpurun_out/b_scale.log | head -40
timeout 900 python -m pytest tests/test_precise_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -rP 2>&1 | grep -v "^$" | tail -150 > gpurun_out/b_precise.log
grep -n "max-abs\|passed\|failed\|Error\|assert" gpurun_out/b_precise.log | head -60
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b_prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic none > $R/gpurun_out/b_bench_c3_prof.json 2> $R/gpurun_out/b_prof.err
cd $R
ls gpurun_out/b_prof_c3 | head; find gpurun_out/b_prof_c3 -name "*kernel_stats.csv" | head -2
f=$(find gpurun_out/b_prof_c3 -name "*kernel_stats.csv" | head -1); head -30 "$f"
find gpurun_out/b_prof_c3 -name "*kernel_trace.csv" -delete; find gpurun_out/b_prof_c3 -name "*.db" -delete

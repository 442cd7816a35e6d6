#!/bin/bash
# round-2 GPU call A: full -m gpu suite, first 32x32x16 A/B, c3 bench with live traffic
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rA 2>&1 | tail -150 > gpurun_out/a_pytest.log
tail -5 gpurun_out/a_pytest.log
timeout 300 python tools/ab_lib.py run base,m32b > gpurun_out/a_ab_m32b.jsonl 2> gpurun_out/a_ab_m32b.err
cat gpurun_out/a_ab_m32b.jsonl; tail -3 gpurun_out/a_ab_m32b.err
timeout 900 python bench.py > gpurun_out/a_bench_c3.json 2> gpurun_out/a_bench_c3.err
tail -c 3000 gpurun_out/a_bench_c3.json; tail -5 gpurun_out/a_bench_c3.err

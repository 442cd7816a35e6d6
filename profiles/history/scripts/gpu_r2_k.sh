#!/bin/bash
# PMC counters of the final build's dominant kernels (persistent 4-wave)
cd ${GRAFT_REPO_ROOT:-.}
bash tools/pmc_gemm.sh v10_swiglu_t197 2688 27648 5120 tile 197 2 > /dev/null 2>&1
bash tools/pmc_gemm.sh v10_qkv_t198 2688 15360 5120 tile 198 0 > /dev/null 2>&1
bash tools/pmc_gemm.sh v10_down_t198 2688 5120 13824 tile 198 0 > /dev/null 2>&1
bash tools/pmc_gemm.sh v10_fc1_t197 32768 4096 1024 tile 197 1 > /dev/null 2>&1
bash tools/pmc_gemm.sh v10_sq_t197 8192 8192 8192 tile 197 0 > /dev/null 2>&1
for t in v10_swiglu_t197 v10_qkv_t198 v10_down_t198 v10_fc1_t197 v10_sq_t197; do echo "== $t"; cat gpurun_out/pmc_$t/summary.txt; done
find gpurun_out/pmc_v10_* -name "*.csv" -delete; find gpurun_out/pmc_v10_* -name "*.db" -delete

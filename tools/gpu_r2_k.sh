#!/bin/bash
# PMC counters of the persistent 4-wave kernel
cd ${GRAFT_REPO_ROOT:-.}
bash tools/pmc_gemm.sh k_sq_t197 8192 8192 8192 tile 197 0 > /dev/null 2>&1
bash tools/pmc_gemm.sh k_swiglu_t197 2688 27648 5120 tile 197 2 > /dev/null 2>&1
bash tools/pmc_gemm.sh k_fc1_t197 32768 4096 1024 tile 197 1 > /dev/null 2>&1
for t in k_sq_t197 k_swiglu_t197 k_fc1_t197; do echo "== $t"; cat gpurun_out/pmc_$t/summary.txt; done
find gpurun_out/pmc_k_* -name "*.csv" -delete; find gpurun_out/pmc_k_* -name "*.db" -delete

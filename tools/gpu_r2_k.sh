#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
bash tools/pmc_gemm.sh k_sq_t100 8192 8192 8192 tile 100 0 > /dev/null 2>&1
bash tools/pmc_gemm.sh k_sq_t99b 8192 8192 8192 tile 99 0 > /dev/null 2>&1
for t in k_sq_t100 k_sq_t99b; do echo "== $t"; cat gpurun_out/pmc_$t/summary.txt; done
find gpurun_out/pmc_k_* -name "*.csv" -delete; find gpurun_out/pmc_k_* -name "*.db" -delete

#!/bin/bash
# PMC counters of the final build's dominant kernels (one rocprofv3 pass per counter group, kernel-trace only)
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
bash tools/pmc_gemm.sh r2_c3_swiglu_t9 2688 27648 5120 tile 9 2 > /dev/null 2>&1
bash tools/pmc_gemm.sh r2_c3_qkv_t96 2688 15360 5120 tile 96 0 > /dev/null 2>&1
bash tools/pmc_gemm.sh r2_c3_qkv_t9 2688 15360 5120 tile 9 0 > /dev/null 2>&1
bash tools/pmc_gemm.sh r2_vit_fc1_t9 32768 4096 1024 tile 9 1 > /dev/null 2>&1
for t in r2_c3_swiglu_t9 r2_c3_qkv_t96 r2_c3_qkv_t9 r2_vit_fc1_t9; do echo "== $t"; cat gpurun_out/pmc_$t/summary.txt; done
find gpurun_out/pmc_r2_* -name "*.csv" -delete; find gpurun_out/pmc_r2_* -name "*.db" -delete

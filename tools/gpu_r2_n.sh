#!/bin/bash
# persistent 4-wave tiles (197 / 198): tests + in-process A/B against 97 / 98 and the shipped tiles
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "gemm" 2>&1 | tail -8
timeout 600 python tools/ab_lib.py run base "32768,4096,1024,1,9|97|197" "32768,1024,4096,0,97|197" "32896,3072,1024,0,9|97|197" "32768,1024,1024,0,9|97|197" \
   "2688,15360,5120,0,98|198|197" "2688,5120,13824,0,98|198" "2688,5120,5120,0,98|198" "2688,27648,5120,2,97|197|198" "8192,8192,8192,0,97|197" "65536,4096,1024,1,9|197" > gpurun_out/n_ab_p4.jsonl 2> gpurun_out/n_err.txt
cat gpurun_out/n_ab_p4.jsonl; tail -3 gpurun_out/n_err.txt

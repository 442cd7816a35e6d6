#!/bin/bash
# 192 x 256 4-wave tiles (99 / 199): tests + in-process A/B on the 13B shapes
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "gemm" 2>&1 | tail -4
timeout 600 python tools/ab_lib.py run base "2688,27648,5120,2,197|199|198" "2688,15360,5120,0,198|199" "2688,5120,13824,0,198|199" "2688,5120,5120,0,198|199" \
   "2688,32008,5120,0,198|199|197" "2816,27648,5120,2,197|199|198" "2816,15360,5120,0,197|97|198" "2816,5120,13824,0,197|97|198" "32896,3072,1024,0,198|199|197" "32768,4096,1024,1,197|199" > gpurun_out/n_ab_192.jsonl 2> gpurun_out/n_err.txt
cat gpurun_out/n_ab_192.jsonl; tail -3 gpurun_out/n_err.txt

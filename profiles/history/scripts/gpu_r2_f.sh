#!/bin/bash
# round-2 GPU call F: rest of the suite after the fixed test; N-split probe for the M = 2688 Llama GEMMs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | grep -v "^$" | tail -12 > gpurun_out/f_pytest.log
tail -4 gpurun_out/f_pytest.log
timeout 600 python tools/ab_lib.py run base "2688,27648,5120,2,51|9|1" "2688,23808,5120,2,51|9|1" "2688,3840,5120,2,8|7|2|3|4|86|76|94" \
   "2688,15360,5120,0,9|51|1" "2688,11776,5120,0,9|51|1" "2688,3584,5120,0,8|7|2|3|4|86|76|94|93" "2688,5120,13824,0,9|51|1" "2688,5120,5120,0,9|51" \
   "1312,22016,4096,2,8|86" "1312,16384,4096,2,8|86|9|51" "1312,12288,4096,0,86|76|9" > gpurun_out/f_nsplit_probe.jsonl 2> gpurun_out/f_err.txt
cat gpurun_out/f_nsplit_probe.jsonl; tail -3 gpurun_out/f_err.txt

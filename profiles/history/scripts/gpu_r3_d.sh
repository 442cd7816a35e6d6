#!/bin/bash
# round-3 GPU call D: full suite (both libraries), PMC of the ViT attention kernel, q|k|v row split
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
O=gpurun_out/d
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider 2>&1 | grep -v "^$" > $O/pytest.log
tail -6 $O/pytest.log
VLY_VIT_ATTN=1 bash tools/pmc_cmd.sh vitattn_v1 vit_attn python tools/vit_attn_time.py 128 > /dev/null 2>&1; cp gpurun_out/pmc_vitattn_v1/summary.txt $O/pmc_vit_attn_v1.txt
bash tools/pmc_cmd.sh vitattn_v2 vit_attn python tools/vit_attn_time.py 128 > /dev/null 2>&1; cp gpurun_out/pmc_vitattn_v2/summary.txt $O/pmc_vit_attn_v2.txt
cat $O/pmc_vit_attn_v1.txt; cat $O/pmc_vit_attn_v2.txt
bash tools/pmc_cmd.sh llattn llama_attn python tools/attn_one.py llama 8 336 40 > /dev/null 2>&1; cp gpurun_out/pmc_llattn/summary.txt $O/pmc_llama_attn.txt; cat $O/pmc_llama_attn.txt
timeout 300 python tools/vit_time.py 128 > $O/vit_128.json 2>> $O/vit_err.txt
VALLEY_VIT_QKV_SPLIT=1 timeout 300 python tools/vit_time.py 128 > $O/vit_128_qkvsplit.json 2>> $O/vit_err.txt
cat $O/vit_128*.json; tail -3 $O/vit_err.txt
rm -rf gpurun_out/pmc_vitattn_v1 gpurun_out/pmc_vitattn_v2 gpurun_out/pmc_llattn

#!/bin/bash
# round-3 GPU call F: attention kernels with un-fused V^T fragment reads
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/f
mkdir -p $O
python tools/ab_lib.py run-vit-attn attnold,base 128 32 256 > $O/ab_vit_attn.jsonl 2> $O/err.txt
VLY_VIT_ATTN=2 python tools/ab_lib.py run-vit-attn attnold,base 128 >> $O/ab_vit_attn.jsonl 2>> $O/err.txt
VLY_VIT_ATTN=2 VLY_VIT_SKEW=4 python tools/ab_lib.py run-vit-attn attnold,base 128 >> $O/ab_vit_attn.jsonl 2>> $O/err.txt
python tools/ab_lib.py run-attn attnold,base 8,336,40 4,328,32 8,352,40 > $O/ab_llama_attn.jsonl 2>> $O/err.txt
cat $O/ab_vit_attn.jsonl $O/ab_llama_attn.jsonl; tail -3 $O/err.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "attention or attn or vit or tower or golden or decode" --timeout 600 -p no:cacheprovider 2>&1 | tail -3

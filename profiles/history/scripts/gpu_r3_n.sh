#!/bin/bash
# round-3 GPU call N: ViT attention with two query tiles per wave and an online softmax (v5)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/n
mkdir -p $O
for F in 4 128 32 256; do
timeout 120 python tools/vit_attn_time.py $F >> $O/vit_attn_v5.jsonl 2>> $O/err.txt
VLY_VIT_ATTN=5 timeout 120 python tools/vit_attn_time.py $F >> $O/vit_attn_v5.jsonl 2>> $O/err.txt
done
cat $O/vit_attn_v5.jsonl; tail -5 $O/err.txt

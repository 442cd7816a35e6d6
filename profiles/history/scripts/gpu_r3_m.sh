#!/bin/bash
# round-3 GPU call M: persistent double-buffered ViT attention (v4)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/m
mkdir -p $O
for F in 4 128 32 256; do
timeout 120 python tools/vit_attn_time.py $F >> $O/vit_attn_v4.jsonl 2>> $O/err.txt
VLY_VIT_ATTN=4 timeout 120 python tools/vit_attn_time.py $F >> $O/vit_attn_v4.jsonl 2>> $O/err.txt
done
cat $O/vit_attn_v4.jsonl; tail -5 $O/err.txt

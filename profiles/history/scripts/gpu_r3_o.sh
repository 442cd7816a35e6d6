#!/bin/bash
# round-3 GPU call O: ViT attention timing variants 4-7 (L2-resident inputs / no HBM traffic / no exp2 / no softmax)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/o
mkdir -p $O
python tools/ab_lib.py run-vit-attn base,vdbg2,vdbg3,vdbg4,vdbg5,vdbg6,vdbg7 128 32 > $O/vit_attn_dbg2.jsonl 2> $O/err.txt
cat $O/vit_attn_dbg2.jsonl; tail -3 $O/err.txt

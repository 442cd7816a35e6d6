#!/bin/bash
# round-3 GPU call K: where does the ViT attention kernel's time go (timing variants, WRONG results by construction)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/k
mkdir -p $O
python tools/ab_lib.py run-vit-attn base,vdbg1,vdbg2,vdbg3 128 32 > $O/vit_attn_dbg.jsonl 2> $O/err.txt
cat $O/vit_attn_dbg.jsonl; tail -3 $O/err.txt

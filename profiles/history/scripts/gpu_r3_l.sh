#!/bin/bash
# round-3 GPU call L: ViT attention with 32-byte output runs (lane transpose) vs 8-byte pieces
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/l
mkdir -p $O
python tools/ab_lib.py run-vit-attn vst0,base 128 32 256 > $O/ab_vit_store.jsonl 2> $O/err.txt
cat $O/ab_vit_store.jsonl; tail -2 $O/err.txt
timeout 120 python tools/vit_attn_time.py 128 2>> $O/err.txt
VLY_VIT_ATTN=3 timeout 120 python tools/vit_attn_time.py 128 2>> $O/err.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "vit or tower or forward_vs_golden" --timeout 600 -p no:cacheprovider 2>&1 | tail -3

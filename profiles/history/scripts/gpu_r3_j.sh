#!/bin/bash
# round-3 GPU call J: ViT attention v3 (257th query split over the keys)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/j
mkdir -p $O
for F in 128 32 256; do
timeout 120 python tools/vit_attn_time.py $F >> $O/vit_attn_v3.jsonl 2>> $O/err.txt
VLY_VIT_ATTN=3 timeout 120 python tools/vit_attn_time.py $F >> $O/vit_attn_v3.jsonl 2>> $O/err.txt
done
cat $O/vit_attn_v3.jsonl; tail -3 $O/err.txt
VLY_VIT_ATTN=3 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_scale_gpu.py -m gpu -q -x -k "vit or tower or forward_vs_golden or two_stream" --timeout 600 -p no:cacheprovider 2>&1 | tail -3
VLY_VIT_ATTN=3 timeout 300 python tools/vit_time.py 128 2>> $O/err.txt
timeout 300 python tools/vit_time.py 128 2>> $O/err.txt

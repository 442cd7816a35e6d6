#!/bin/bash
# round-3 GPU call B: tr-read probe, ViT attention v1/v2, stagger A/B, full-depth parity tests + oracle tool
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
O=gpurun_out/b
mkdir -p $O
./tools/probes/tr_read > $O/tr_read.txt 2>&1; head -20 $O/tr_read.txt
VLY_VIT_ATTN=1 timeout 120 python tools/vit_attn_time.py 128 > $O/vit_attn.jsonl 2> $O/err.txt
timeout 120 python tools/vit_attn_time.py 128 >> $O/vit_attn.jsonl 2>> $O/err.txt
VLY_VIT_ATTN=1 timeout 120 python tools/vit_attn_time.py 32 >> $O/vit_attn.jsonl 2>> $O/err.txt
timeout 120 python tools/vit_attn_time.py 32 >> $O/vit_attn.jsonl 2>> $O/err.txt
cat $O/vit_attn.jsonl; tail -3 $O/err.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "vit or tower or forward_vs_golden" --timeout 600 -p no:cacheprovider 2>&1 | tail -4
timeout 600 python tools/ab_lib.py run base,stag4,stag8,stag16 32768,4096,1024,1,197 32896,3072,1024,0,198 32768,1024,1024,0,197 32768,1024,4096,0,197 \
  2688,27648,5120,2,197 2688,15360,5120,0,198 > $O/ab_stagger.jsonl 2> $O/ab_err.txt
cat $O/ab_stagger.jsonl; tail -3 $O/ab_err.txt
timeout 900 python -m pytest tests/test_depth_gpu.py -m gpu -q -x -s --timeout 800 -p no:cacheprovider 2>&1 | grep -v "^$" > $O/depth_tests.log
tail -25 $O/depth_tests.log
timeout 600 python tools/full_depth_oracle.py > $O/full_depth_oracle_7b.json 2> $O/full_depth_err.txt
cat $O/full_depth_oracle_7b.json; tail -5 $O/full_depth_err.txt

#!/bin/bash
# round-3 GPU call C: full suite after the dtype-neutral refactor, fp16 library tests, depth tests, ViT attention skew sweep,
# fp16 / fp32 bench price tags
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
O=gpurun_out/c
mkdir -p $O
for sk in 0 2 4 8 16 32; do VLY_VIT_SKEW=$sk timeout 120 python tools/vit_attn_time.py 128 2>> $O/err.txt | sed "s/}$/, \"skew\": $sk}/" >> $O/vit_attn_skew.jsonl; done
VLY_VIT_ATTN=1 timeout 120 python tools/vit_attn_time.py 128 >> $O/vit_attn_skew.jsonl 2>> $O/err.txt
cat $O/vit_attn_skew.jsonl; tail -3 $O/err.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 1200 -p no:cacheprovider -rP 2>&1 | grep -v "^$" > $O/pytest.log
tail -5 $O/pytest.log; grep -n "layers:\|decode steps\|quantity\|_rel \|_maxabs \|absmax" $O/pytest.log | head -60
VALLEY_PRECISION=fp16 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --also none > $O/bench_c3_fp16.json 2> $O/bench_fp16_err.txt
head -c 1500 $O/bench_c3_fp16.json; echo; tail -3 $O/bench_fp16_err.txt
VALLEY_PRECISION=fp32 VALLEY_PACK_WEIGHTS=0 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --traffic none --also none --no-kernel-events > $O/bench_c3_fp32.json 2> $O/bench_fp32_err.txt
head -c 1500 $O/bench_c3_fp32.json; echo; tail -3 $O/bench_fp32_err.txt

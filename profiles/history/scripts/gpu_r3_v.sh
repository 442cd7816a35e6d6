#!/bin/bash
# round-3 GPU call V: slices per remainder tile (VLY_P4_SK_S) per shape
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/v3
mkdir -p $O
for S in 2 4 8; do
echo "S=$S" >> $O/sk_s2.jsonl
VLY_P4_SK_S=$S timeout 300 python tools/gemm_ab.py "2688,27648,5120,2:197,s298,s299;2816,27648,5120,2:197,s298;1312,22016,4096,2:8,s298,s299;1312,12288,4096,0:86,s298,s299;1312,4096,4096,0:94,s298,s299;1312,4096,11008,0:94,s298,s299" >> $O/sk_s2.jsonl 2>> $O/err.txt
done
cat $O/sk_s2.jsonl; tail -3 $O/err.txt

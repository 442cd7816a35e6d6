#!/bin/bash
# round-3 GPU call P: GEMV rows per workgroup (2 vs 4) on the decode projection shapes (13B then 7B), cold weights
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/p
mkdir -p $O
S="5120,5120,0,1;5120,13824,0,1;4096,4096,0,1;4096,11008,0,1;15360,5120,0,0;27648,5120,2,0"
for R in 2 4 2 4; do
echo "rows $R" >> $O/gemv_rows.jsonl
VLY_GEMV_ROWS=$R timeout 300 python tools/gemv_sweep.py "$S" >> $O/gemv_rows.jsonl 2>> $O/err.txt
done
cat $O/gemv_rows.jsonl; tail -3 $O/err.txt

#!/bin/bash
# round-3 GPU call Z: llama_attn2_kernel (LDS-DMA tiles, two workgroups per CU) — every attention test under it, then timing 1 vs 2
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/z3
mkdir -p $O
VLY_LLAMA_ATTN=2 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "attention or llama or forward or golden" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for V in 1 2 1 2; do
for SH in "8 336 40" "4 328 32" "8 352 40" "2 1024 40"; do
echo -n "attn=$V " >> $O/attn_time.txt
VLY_LLAMA_ATTN=$V timeout 120 python tools/attn_one.py llama $SH >> $O/attn_time.txt 2>> $O/err.txt
done
done
cat $O/attn_time.txt; tail -3 $O/err.txt

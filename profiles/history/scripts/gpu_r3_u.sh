#!/bin/bash
# round-3 GPU call U: the persistent kernel's stream-K form — correctness, then per-shape timing against the plain form
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/u3
mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "p4_streamk" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
grep -q "failed\|rror" $O/pytest.log && exit 1
timeout 600 python tools/gemm_ab.py "2688,27648,5120,2:197,s297,s299;2688,15360,5120,0:198,s298,s297;2688,5120,13824,0:198,s298,s297;2688,5120,5120,0:198,s298,s297;1312,22016,4096,2:8,197,s297,s299;32896,3072,1024,0:198,s298" > $O/gemm_time.jsonl 2>> $O/err.txt
cat $O/gemm_time.jsonl; tail -3 $O/err.txt

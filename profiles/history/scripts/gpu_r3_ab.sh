#!/bin/bash
# round-3 GPU call AB: one rank's share of configs[3] as an 8-rank job with the prefill replicated (B = 64, M = 22528), simulated at N = 1
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/ab3
mkdir -p $O
VALLEY_TUNE_CACHE=$O/tune_c4_rep.json timeout 900 python bench.py --config c4 --prefill replicated --simulate-ranks 8 --steps 5 --warmup 1 --no-cpu-baseline --traffic none --also none > $O/c4_replicated_sim8.json 2> $O/err.txt
head -c 1800 $O/c4_replicated_sim8.json; echo; tail -3 $O/err.txt

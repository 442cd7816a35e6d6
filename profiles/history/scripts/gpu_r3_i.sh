#!/bin/bash
# round-3 GPU call I: tuner decisions for the row-split shapes at 32 / 64 frames (merged into valley_amd/tuned/gfx950.json)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/i
mkdir -p $O
for F in 32 64; do
VALLEY_TUNE_CACHE=$PWD/$O/tune_new.json timeout 600 python tools/vit_time.py $F >> $O/vit.jsonl 2>> $O/err.txt
done
cat $O/vit.jsonl; tail -2 $O/err.txt; python -c "
import json; d=json.load(open('$O/tune_new.json')); print(len(d)); [print(e) for e in d]"

#!/bin/bash
# round-3 GPU call AH: c5 decode, the norm-into-LDS form (previous commit's gemv_bf16.hip as a variant library) vs the factored form
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/ah3
mkdir -p $O
for V in prev new prev new; do
if [ $V = prev ]; then export VALLEY_HIP_LIB=$PWD/valley_amd/lib/variants/libvalley_hip_gemvprev.so; else unset VALLEY_HIP_LIB; fi
timeout 300 python bench.py --config c5 --decode 256 --warmup 8 --also none > $O/dec_$V.json 2>> $O/err.txt
python - <<PY
import json
ln=[l for l in open("$O/dec_$V.json") if l.startswith("{")]
j=json.loads(ln[-1]); print("$V", j["value"], j["ms_per_step"])
PY
done
tail -2 $O/err.txt

#!/bin/bash
# round-3 GPU call AF: the fp16 library end to end on the final build (c3 short, c2 short, decode)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/af3
mkdir -p $O
for A in "--config c3" "--config c2" "--config c5 --decode 64"; do
VALLEY_PRECISION=fp16 timeout 600 python bench.py $A --steps 8 --warmup 3 --no-cpu-baseline --traffic none --also none >> $O/fp16.jsonl 2>> $O/err.txt
done
python - <<'PY'
import json
for l in open("gpurun_out/af3/fp16.jsonl"):
    if l.startswith("{"):
        j = json.loads(l); print(j["config"]["name"], j["dtype"], j["value"], j["unit"], j["ms_per_step"], (j.get("stages") or {}).get("prefill_frac_of_bf16_peak"))
PY
tail -2 $O/err.txt

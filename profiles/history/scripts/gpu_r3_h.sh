#!/bin/bash
# round-3 GPU call H: two-stream remainder schedule
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/h
mkdir -p $O
timeout 600 python -m pytest tests/test_scale_gpu.py -m gpu -q -x -k two_stream --timeout 600 -p no:cacheprovider 2>&1 | tail -4
for F in 128 256; do
timeout 300 python tools/vit_time.py $F >> $O/vit_2s.jsonl 2>> $O/err.txt
VALLEY_VIT_TWO_STREAM=1 timeout 300 python tools/vit_time.py $F >> $O/vit_2s.jsonl 2>> $O/err.txt
done
for F in 32 64; do
VALLEY_ROW_SPLIT_MIN=8192 timeout 300 python tools/vit_time.py $F >> $O/vit_2s.jsonl 2>> $O/err.txt
VALLEY_ROW_SPLIT_MIN=8192 VALLEY_VIT_TWO_STREAM=1 timeout 300 python tools/vit_time.py $F >> $O/vit_2s.jsonl 2>> $O/err.txt
done
cat $O/vit_2s.jsonl; tail -2 $O/err.txt
VALLEY_VIT_TWO_STREAM=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --also none > $O/bench_c3_2s.json 2>> $O/err.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --also none > $O/bench_c3_1s.json 2>> $O/err.txt
python - <<'PY'
import json
for f in ("gpurun_out/h/bench_c3_1s.json", "gpurun_out/h/bench_c3_2s.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["stages"]["vit_ms"], d["stages"]["prefill_ms"])
PY

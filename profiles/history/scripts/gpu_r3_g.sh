#!/bin/bash
# round-3 GPU call G: row split at smaller F (c2's ViT batch), hidden-states test, c2 bench with the split
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/g
mkdir -p $O
timeout 300 python -m pytest tests/test_precise_gpu.py -m gpu -q -x -s -k hidden_states --timeout 300 -p no:cacheprovider 2>&1 | grep -v "^$" | tail -6
for F in 32 64; do
timeout 300 python tools/vit_time.py $F >> $O/vit_split.jsonl 2>> $O/err.txt
VALLEY_ROW_SPLIT_MIN=8192 timeout 300 python tools/vit_time.py $F >> $O/vit_split.jsonl 2>> $O/err.txt
done
cat $O/vit_split.jsonl; tail -2 $O/err.txt
timeout 300 python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --traffic none --also none > $O/bench_c2.json 2>> $O/err.txt
VALLEY_ROW_SPLIT_MIN=8192 timeout 300 python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --traffic none --also none > $O/bench_c2_split.json 2>> $O/err.txt
python - <<'PY'
import json
for f in ("gpurun_out/g/bench_c2.json", "gpurun_out/g/bench_c2_split.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["stages"]["vit_ms"], d["stages"]["prefill_ms"], d["config"]["tune_passes"])
    for k, v in d["roofline"]["gemm_shapes"].items():
        if k.startswith("82") or k.startswith("81") or k.startswith("32x"):
            print("   ", k, v)
PY

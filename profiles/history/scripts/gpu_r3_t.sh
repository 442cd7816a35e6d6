#!/bin/bash
# round-3 GPU call T: gate|up cut at 2304 rows (ops.round_split): tuned decisions of the two new shapes, then c3 off / on interleaved
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/t3
mkdir -p $O
VALLEY_TUNE_CACHE=$O/tune_c3.json timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --also none > $O/c3_tune.json 2>> $O/err.txt
for F in 0 1 0 1; do
VALLEY_TUNE_CACHE=$O/tune_c3.json VALLEY_ROUND_SPLIT=$F timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --also none > $O/c3_split$F.json 2>> $O/err.txt
python - <<PY
import json
ln=[l for l in open("$O/c3_split$F.json") if l.startswith("{")]
j=json.loads(ln[-1]); st=j["stages"]; r=j["roofline"]
print("round_split=$F", j["value"], j["ms_per_step"], "vit", st["vit_ms"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"], "tune_passes", j["config"]["tune_passes"])
print("   ", {k: (v["TFLOPs"], v["avg_us"], v["kernel"]) for k, v in r["gemm_shapes"].items() if "27648" in k})
PY
done
python - <<'PY'
import json
for e in json.load(open("gpurun_out/t3/tune_c3.json")):
    if e["key"][0] in (2304, 384): print(json.dumps(e))
PY
tail -3 $O/err.txt

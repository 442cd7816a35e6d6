#!/bin/bash
# round-3 GPU call X: c3 and c4 re-tuned from scratch with the split-K remainder candidates vs the shipped table
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/x3
mkdir -p $O
for C in c3 c4; do
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$O/tune_$C.json timeout 1200 python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline --traffic none --also none > $O/${C}_retune.json 2>> $O/err.txt
timeout 600 python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --traffic none --also none > $O/${C}_shipped.json 2>> $O/err.txt
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$O/tune_$C.json timeout 600 python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --traffic none --also none > $O/${C}_retuned.json 2>> $O/err.txt
done
python - <<'PY'
import json
for f in ("c3_shipped", "c3_retuned", "c4_shipped", "c4_retuned"):
    ln=[l for l in open(f"gpurun_out/x3/{f}.json") if l.startswith("{")]
    j=json.loads(ln[-1]); st=j["stages"]
    print(f, j["value"], j["ms_per_step"], "vit", st["vit_ms"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"], "tune_passes", j["config"]["tune_passes"])
    print("   ", {k: (v["TFLOPs"], v["avg_us"], v["kernel"][:34]) for k, v in j["roofline"]["gemm_shapes"].items() if v["ms_per_step"] > 1.0})
PY
tail -3 $O/err.txt

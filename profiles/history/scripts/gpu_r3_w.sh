#!/bin/bash
# round-3 GPU call W: split-K remainder kernels — tests, then c2 and c3 through the online tuner (new decisions saved)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/w3
mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "p4_streamk" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
grep -q "failed\|rror" $O/pytest.log && exit 1
# re-tune the c2 / c3 prefill shapes from scratch (table off for them is not possible per shape: VALLEY_TUNE_TABLE=0 re-tunes all)
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$O/tune_c2.json timeout 900 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --traffic none --also none > $O/c2_retuned.json 2>> $O/err.txt
timeout 600 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --traffic none --also none > $O/c2_shipped.json 2>> $O/err.txt
VALLEY_TUNE_CACHE=$O/tune_c2.json timeout 600 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --traffic none --also none > $O/c2_retuned2.json 2>> $O/err.txt
python - <<'PY'
import json
for f in ("c2_shipped", "c2_retuned", "c2_retuned2"):
    ln=[l for l in open(f"gpurun_out/w3/{f}.json") if l.startswith("{")]
    j=json.loads(ln[-1]); st=j["stages"]
    print(f, j["value"], j["ms_per_step"], "vit", st["vit_ms"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"], "tune_passes", j["config"]["tune_passes"])
    print("   ", {k: (v["TFLOPs"], v["avg_us"], v["kernel"]) for k, v in j["roofline"]["gemm_shapes"].items() if k.startswith("1312")})
PY
tail -3 $O/err.txt

#!/bin/bash
# final-build numbers of the other configs: c2 (re-tuned with 99/199 among the candidates), c4, decode
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
B="python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3"
timeout 900 $B --config c2 > gpurun_out/r_c2_shipped.json 2> gpurun_out/r_err1.txt
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$R/gpurun_out/r_tune_c2.json timeout 1500 $B --config c2 > gpurun_out/r_c2_retuned.json 2> gpurun_out/r_err2.txt
timeout 900 $B --config c4 > gpurun_out/r_c4.json 2> gpurun_out/r_err3.txt
timeout 900 python bench.py --no-cpu-baseline --traffic none --decode 256 > gpurun_out/r_decode13b.json 2> gpurun_out/r_err4.txt
python - <<'PY'
import json
for f in ("c2_shipped", "c2_retuned", "c4"):
    try:
        j = json.load(open(f"gpurun_out/r_{f}.json"))
        st = j["stages"]
        print(f, j["value"], "ms", j["ms_per_step"], "vit", st["vit_ms"], st["vit_frames_per_s_per_gpu"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_tokens_per_s_per_gpu"], st["prefill_frac_of_bf16_peak"],
              {k: (v["TFLOPs"], v["kernel"][:32]) for k, v in list(j["roofline"]["gemm_shapes"].items())[:9]})
    except Exception as e:
        print(f, "FAILED", e)
try:
    j = json.load(open("gpurun_out/r_decode13b.json")); print("decode", j["value"], j["unit"], j.get("roofline", {}).get("achieved"), j.get("roofline", {}).get("frac"))
except Exception as e:
    print("decode FAILED", e)
PY

#!/bin/bash
# round-2 GPU call H: 224-row tiles — kernel tests, in-process A/B on the M = 2688 shapes, re-tuned c3 bench
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gemm" 2>&1 | tail -4
timeout 600 python tools/ab_lib.py run base "2688,15360,5120,0,9|1|51|95|96" "2688,5120,13824,0,9|51|95|96" "2688,5120,5120,0,9|51|95|96" "2688,27648,5120,2,51|1|95|96" \
   "1312,22016,4096,2,8|95|96" "1312,12288,4096,0,76|95|96" > gpurun_out/h_ab_224.jsonl 2> gpurun_out/h_err.txt
cat gpurun_out/h_ab_224.jsonl; tail -2 gpurun_out/h_err.txt
B="python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3"
timeout 900 $B > gpurun_out/h_c3_shipped_table.json 2> gpurun_out/h_err1.txt
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$R/gpurun_out/h_tune_full.json timeout 1500 $B > gpurun_out/h_c3_retuned.json 2> gpurun_out/h_err2.txt
python - <<'PY'
import json
for f in ("c3_shipped_table", "c3_retuned"):
    try:
        j = json.load(open(f"gpurun_out/h_{f}.json"))
        st = j["stages"]
        print(f, j["value"], "ms", j["ms_per_step"], "vit", st["vit_ms"], st["vit_frames_per_s_per_gpu"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"],
              "tune", j["config"]["tune_passes"], {k: (v["TFLOPs"], v["avg_us"], v["kernel"][12:40]) for k, v in list(j["roofline"]["gemm_shapes"].items())[:9]})
    except Exception as e:
        print(f, "FAILED", e)
PY

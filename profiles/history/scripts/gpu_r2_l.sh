#!/bin/bash
# round-2 GPU call L: the 4-wave tiles (97 / 98) — kernel tests, in-process A/B on the hot shapes, re-tuned benches
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gemm" 2>&1 | tail -4
timeout 600 python tools/ab_lib.py run base "2688,15360,5120,0,96|97|98" "2688,5120,13824,0,9|95|97|98" "2688,5120,5120,0,9|96|97|98" "2688,27648,5120,2,9|1|97|98" \
   "32768,4096,1024,1,9|97" "32768,1024,4096,0,1|9|97" "32896,3072,1024,0,9|97" "1312,22016,4096,2,8|97|98" "1312,12288,4096,0,76|97|98" "8192,8192,8192,0,9|97" > gpurun_out/l_ab_4wave.jsonl 2> gpurun_out/l_err.txt
cat gpurun_out/l_ab_4wave.jsonl; tail -2 gpurun_out/l_err.txt
B="python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3"
timeout 900 $B > gpurun_out/l_c3_shipped_table.json 2> gpurun_out/l_err1.txt
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$R/gpurun_out/l_tune_c3.json timeout 1500 $B > gpurun_out/l_c3_retuned.json 2> gpurun_out/l_err2.txt
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$R/gpurun_out/l_tune_c2.json timeout 1500 $B --config c2 > gpurun_out/l_c2_retuned.json 2> gpurun_out/l_err3.txt
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$R/gpurun_out/l_tune_c4.json timeout 1500 $B --config c4 > gpurun_out/l_c4_retuned.json 2> gpurun_out/l_err4.txt
python - <<'PY'
import json
for f in ("c3_shipped_table", "c3_retuned", "c2_retuned", "c4_retuned"):
    try:
        j = json.load(open(f"gpurun_out/l_{f}.json"))
        st = j["stages"]
        print(f, j["value"], "ms", j["ms_per_step"], "vit", st["vit_ms"], st["vit_frames_per_s_per_gpu"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"],
              "tune", j["config"]["tune_passes"], {k: (v["TFLOPs"], v["avg_us"], v["kernel"][12:40]) for k, v in list(j["roofline"]["gemm_shapes"].items())[:9]})
    except Exception as e:
        print(f, "FAILED", e)
PY

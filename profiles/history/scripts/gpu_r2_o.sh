#!/bin/bash
# re-tuned benches with the persistent tiles among the candidates
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
B="python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3"
timeout 900 $B > gpurun_out/o_c3_shipped_table.json 2> gpurun_out/o_err1.txt
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$R/gpurun_out/o_tune_c3.json timeout 1500 $B > gpurun_out/o_c3_retuned.json 2> gpurun_out/o_err2.txt
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$R/gpurun_out/o_tune_c2.json timeout 1500 $B --config c2 > gpurun_out/o_c2_retuned.json 2> gpurun_out/o_err3.txt
VALLEY_TUNE_TABLE=0 VALLEY_TUNE_CACHE=$R/gpurun_out/o_tune_c4.json timeout 1500 $B --config c4 > gpurun_out/o_c4_retuned.json 2> gpurun_out/o_err4.txt
python - <<'PY'
import json
for f in ("c3_shipped_table", "c3_retuned", "c2_retuned", "c4_retuned"):
    try:
        j = json.load(open(f"gpurun_out/o_{f}.json"))
        st = j["stages"]
        print(f, j["value"], "ms", j["ms_per_step"], "vit", st["vit_ms"], st["vit_frames_per_s_per_gpu"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"],
              "tune", j["config"]["tune_passes"], {k: (v["TFLOPs"], v["avg_us"], v["kernel"][:34]) for k, v in list(j["roofline"]["gemm_shapes"].items())[:9]})
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/o_err2.txt

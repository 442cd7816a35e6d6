#!/bin/bash
# fused RoPE epilogue on the persistent tiles: c3 A/B on one box
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
B="python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3"
timeout 900 $B > gpurun_out/q_c3_unfused.json 2> gpurun_out/q_err1.txt
VALLEY_FUSE_ROPE=1 VALLEY_TUNE_CACHE=$R/gpurun_out/q_tune_rope.json timeout 900 $B > gpurun_out/q_c3_fused.json 2> gpurun_out/q_err2.txt
timeout 900 $B > gpurun_out/q_c3_unfused2.json 2> gpurun_out/q_err3.txt
VALLEY_FUSE_ROPE=1 VALLEY_TUNE_CACHE=$R/gpurun_out/q_tune_rope.json timeout 900 $B > gpurun_out/q_c3_fused2.json 2> gpurun_out/q_err4.txt
python - <<'PY'
import json
for f in ("c3_unfused", "c3_fused", "c3_unfused2", "c3_fused2"):
    try:
        j = json.load(open(f"gpurun_out/q_{f}.json"))
        st = j["stages"]
        print(f, j["value"], "ms", j["ms_per_step"], "prefill", st["prefill_ms"], "tune", j["config"]["tune_passes"],
              {k: (v["TFLOPs"], v["avg_us"], v["kernel"][:30]) for k, v in list(j["roofline"]["gemm_shapes"].items())[:9] if "15360" in k})
    except Exception as e:
        print(f, "FAILED", e)
PY
cat gpurun_out/q_tune_rope.json | tr -d '\n' | head -c 600

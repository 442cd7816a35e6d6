#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "gemm" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3"
timeout 900 $B > gpurun_out/q_c3.json 2> gpurun_out/q_err1.txt
timeout 900 $B --config c4 > gpurun_out/q_c4.json 2> gpurun_out/q_err2.txt
python - <<'PY'
import json
for f in ("c3", "c4"):
    try:
        j = json.load(open(f"gpurun_out/q_{f}.json"))
        st = j["stages"]
        print(f, j["value"], "ms", j["ms_per_step"], "vit", st["vit_ms"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"],
              {k: (v["TFLOPs"], v["kernel"][:30]) for k, v in list(j["roofline"]["gemm_shapes"].items())[:9]})
    except Exception as e:
        print(f, "FAILED", e)
PY

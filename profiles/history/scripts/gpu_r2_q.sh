#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "gemm" 2>&1 | tail -2
VALLEY_GEMM_MODE=tiles timeout 900 python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3 > gpurun_out/q_c3_tiles.json 2> gpurun_out/q_err1.txt
timeout 900 python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3 > gpurun_out/q_c3.json 2> gpurun_out/q_err2.txt
python - <<'PY'
import json
for f in ("c3_tiles", "c3"):
    j = json.load(open(f"gpurun_out/q_{f}.json")); st = j["stages"]
    print(f, j["value"], "ms", j["ms_per_step"], "vit", st["vit_ms"], "prefill", st["prefill_ms"], {k: (v["TFLOPs"], v["kernel"][:30]) for k, v in list(j["roofline"]["gemm_shapes"].items())[:9]})
PY

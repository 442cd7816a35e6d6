#!/bin/bash
# round-3 GPU call AD: FUSE_ROPE=auto — model / scale / depth tests and the c3 line with the shipped table
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/ad3
mkdir -p $O
timeout 900 python -m pytest tests/test_scale_gpu.py tests/test_depth_gpu.py tests/test_model_gpu.py tests/test_precise_gpu.py -q -x -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --also none > $O/c3.json 2>> $O/err.txt
python - <<'PY'
import json
ln=[l for l in open("gpurun_out/ad3/c3.json") if l.startswith("{")]
j=json.loads(ln[-1]); st=j["stages"]
print(j["value"], j["ms_per_step"], "vit", st["vit_ms"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"], "tune_passes", j["config"]["tune_passes"])
print({k: (v["TFLOPs"], v["avg_us"]) for k, v in j["roofline"]["gemm_shapes"].items() if "15360" in k})
PY
tail -2 $O/err.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_scale_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3 > gpurun_out/q_c3.json 2> gpurun_out/q_err1.txt
python - <<'PY'
import json
j = json.load(open("gpurun_out/q_c3.json")); st = j["stages"]
print("c3", j["value"], "ms", j["ms_per_step"], "vit", st["vit_ms"], "prefill", st["prefill_ms"], j["roofline"]["kernel"], j["roofline"]["achieved"])
PY

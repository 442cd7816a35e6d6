#!/bin/bash
# round-3 GPU call S: decode attention split over the keys + merge inside the o GEMV — unit tests, then c5 decode off / on
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s3
mkdir -p $O
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -x -k "decode_attention or gemv" > $O/pytest_unit.log 2>&1; tail -5 $O/pytest_unit.log
grep -q "failed\|error" $O/pytest_unit.log && exit 1
for F in 0 1 0 1; do
VALLEY_DECODE_SPLIT_ATTN=$F timeout 600 python bench.py --config c5 --decode 256 --warmup 8 --also none > $O/dec_split${F}.json 2>> $O/err.txt
python - <<PY
import json
ln=[l for l in open("$O/dec_split${F}.json") if l.startswith("{")]
j=json.loads(ln[-1]); print("split_attn=$F", j["value"], j["unit"], j["ms_per_step"], j.get("roofline",{}).get("frac"))
PY
done
timeout 900 python -m pytest tests/test_depth_gpu.py tests/test_scale_gpu.py tests/test_model_gpu.py -q -x -k "decode or graph or batch" > $O/pytest_decode.log 2>&1; tail -3 $O/pytest_decode.log
tail -3 $O/err.txt

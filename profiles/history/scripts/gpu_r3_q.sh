#!/bin/bash
# round-3 GPU call Q: RMSNorm folded into the decode GEMVs — bit-identity test, then c5 decode with the fusion off / on
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/q3
mkdir -p $O
timeout 150 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemv" > $O/pytest_gemv.log 2>&1; tail -3 $O/pytest_gemv.log
for F in 0 1 0 1; do
VALLEY_DECODE_FUSE_NORM=$F timeout 600 python bench.py --config c5 --decode 256 --warmup 8 --also none > $O/dec_fuse${F}.json 2>> $O/err.txt
python - <<PY
import json
ln=[l for l in open("$O/dec_fuse${F}.json") if l.startswith("{")]
j=json.loads(ln[-1]); print("fuse_norm=$F", j["value"], j["unit"], j["ms_per_step"], j.get("roofline",{}).get("frac"))
PY
done
timeout 900 python -m pytest tests/test_depth_gpu.py tests/test_scale_gpu.py -q -x -k "decode or graph" > $O/pytest_decode.log 2>&1; tail -3 $O/pytest_decode.log
tail -3 $O/err.txt

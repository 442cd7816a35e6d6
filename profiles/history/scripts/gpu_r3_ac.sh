#!/bin/bash
# round-3 GPU call AC: the fused RoPE / KV epilogue again, behind the new attention kernel (VALLEY_FUSE_ROPE=0 / 1, interleaved)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/ac3
mkdir -p $O
for F in 0 1 0 1; do
VALLEY_TUNE_CACHE=$O/tune_rope.json VALLEY_FUSE_ROPE=$F timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --also none > $O/c3_rope$F.json 2>> $O/err.txt
python - <<PY
import json
ln=[l for l in open("$O/c3_rope$F.json") if l.startswith("{")]
j=json.loads(ln[-1]); st=j["stages"]
print("fuse_rope=$F", j["value"], j["ms_per_step"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"], "tune_passes", j["config"]["tune_passes"])
PY
done
tail -2 $O/err.txt

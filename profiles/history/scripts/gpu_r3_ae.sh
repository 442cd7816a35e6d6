#!/bin/bash
# round-3 GPU call AE: fused RoPE epilogue on the c4 and c2 shapes (VALLEY_FUSE_ROPE=0 / 1)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/ae3
mkdir -p $O
for C in c4 c2; do
for F in 0 1 0 1; do
VALLEY_TUNE_CACHE=$O/tune_rope_$C.json VALLEY_FUSE_ROPE=$F timeout 600 python bench.py --config $C --steps 15 --warmup 4 --no-cpu-baseline --traffic none --also none > $O/${C}_rope$F.json 2>> $O/err.txt
python - <<PY
import json
ln=[l for l in open("$O/${C}_rope$F.json") if l.startswith("{")]
j=json.loads(ln[-1]); st=j["stages"]
print("$C fuse_rope=$F", j["value"], j["ms_per_step"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"], "tune_passes", j["config"]["tune_passes"])
PY
done
done
python - <<'PY'
import json
for c in ("c4", "c2"):
    for e in json.load(open(f"gpurun_out/ae3/tune_rope_{c}.json")):
        if e["key"][3] == 4: print(json.dumps(e))
PY
tail -2 $O/err.txt

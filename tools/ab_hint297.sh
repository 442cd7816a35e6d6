#!/bin/bash
# In-step A/B of the 256-row split-K remainder (vly_gemm_bf16_streamk tile_hint 297) against the shipped choices for the 13B
# prefill GEMMs of configs[2] (M = 2688): override files for VALLEY_TUNE_CACHE, the default bench per arm.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/ab_297; mkdir -p $O
python - <<'PY'
import json
e = json.load(open("valley_amd/tuned/gfx950.json"))
def over(pred, name):
    out = []
    for x in e:
        k = x["key"]
        if k[0] in (2688, 2816) and len(k) == 8 and pred(k):
            out.append({"key": k, "kind": "sk", "tile": 297})
    json.dump(out, open(f"gpurun_out/ab_297/{name}.json", "w"))
    print(name, len(out))
over(lambda k: k[3] == 2, "gu")
over(lambda k: k[3] == 2 or (k[1] == 15360 and k[3] in (0, 4)), "gu_qkv")
over(lambda k: k[3] in (0, 2, 4), "gu_qkv_all0")
PY
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "p4_streamk" 2>&1 | tail -3 | tee $O/tests.txt
for arm in d gu d gu gu_qkv; do
  if [ $arm = d ]; then unset VALLEY_TUNE_CACHE; else export VALLEY_TUNE_CACHE=$O/$arm.json; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --also none 2>$O/err_$arm.txt | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); g=d['roofline']['gemm_shapes']
print('arm=$arm', d['value'], d['stages']['prefill_ms'], ' '.join('%s:%.0f' % (k, v['TFLOPs']) for k,v in g.items() if k.startswith('2688')))"
done | tee $O/ab.txt
tail -3 $O/err_gu.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/ab_gm
for arm in d 11 d 11; do
  if [ $arm = d ]; then unset VLY_TILE_GM VLY_TILE_GM_MAXM; else export VLY_TILE_GM=$arm VLY_TILE_GM_MAXM=4096; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --also none 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); g=d['roofline']['gemm_shapes']
print('gm=$arm', d['value'], d['stages']['prefill_ms'], ' '.join('%s:%.0f' % (k, v['TFLOPs']) for k,v in g.items() if k.startswith('2688')))"
done | tee gpurun_out/ab_gm/ab.txt

#!/bin/bash
# In-step A/B of the XCD phase skew of the persistent GEMM (VLY_P4_XCD_SKEW_NS [, VLY_P4_XCD_SKEW_MAXK]) on the default bench.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/ab_skew; mkdir -p $O
for arm in ${ARMS:-"0 0" "1000 1024" "2500 1024" "0 0" "1000 99999" "2500 99999"}; do
  set -- $arm
  VLY_P4_XCD_SKEW_NS=$1 VLY_P4_XCD_SKEW_MAXK=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --also none 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); g=d['roofline']['gemm_shapes']
print('skew=$1 maxk=$2', d['value'], 'vit', d['stages']['vit_ms'], 'prefill', d['stages']['prefill_ms'], ' '.join('%s:%.0f' % (k.split('/')[0][-14:], v['TFLOPs']) for k,v in g.items() if v['TFLOPs'] > 900))"
done | tee $O/ab.txt

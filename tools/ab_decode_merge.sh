#!/bin/bash
# Interleaved A/B: where the split attention's partials are merged (VALLEY_DECODE_MERGE = attn | oproj), configs[4].
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/ab_merge; mkdir -p $O
timeout 600 python -m pytest tests/test_decode_merge_gpu.py tests/test_decode_persistent_gpu.py -q -x 2>&1 | tail -3 | tee $O/tests.txt
for arm in oproj attn oproj attn; do
  VALLEY_DECODE_MERGE=$arm timeout 200 python bench.py --config c5 --decode 256 --warmup 8 --no-cpu-baseline --traffic none --also none 2>$O/err_$arm.txt | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('merge=$arm', d['value'], d['unit'], d.get('ms_per_step'))"
done | tee $O/ab.txt
tail -2 $O/err_attn.txt

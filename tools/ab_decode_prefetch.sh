#!/bin/bash
# Interleaved A/B of the decode step's weight prefetch (VALLEY_DECODE_PREFETCH) on configs[4] (13B, 256 tokens).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/ab_prefetch; mkdir -p $O
timeout 300 python -m pytest tests/test_decode_prefetch_gpu.py -q -x 2>&1 | tail -3 | tee $O/tests.txt
for arm in ${ARMS:-0 o 0 o}; do
  VALLEY_DECODE_PREFETCH=$arm timeout 200 python bench.py --config c5 --decode 256 --warmup 8 --no-cpu-baseline --traffic none --also none 2>$O/err_$arm.txt | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('prefetch=$arm', d['value'], d['unit'], d.get('ms_per_step'))"
done | tee $O/ab.txt
tail -2 $O/err_o.txt

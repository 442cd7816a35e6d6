#!/bin/bash
# Interleaved A/B (round 6): split_merge_if_last with relaxed sc1 hand-off (shipped) vs release / acquire fences around the ticket
# (variant mfence = attention.hip -DVLY_DECODE_MERGE_FENCE=1), configs[4] at one and at eight live requests.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for arm in base mfence base mfence; do
  L=valley_amd/lib/libvalley_hip.so; [ $arm = mfence ] && L=valley_amd/lib/variants/libvalley_hip_mfence.so
  for extra in "" "--decode-batch 8"; do
    VALLEY_HIP_LIB=$PWD/$L timeout 300 python bench.py --config c5 --decode 192 --warmup 8 $extra --no-cpu-baseline --traffic none --also none 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$arm', '$extra', d['value'], d['unit'], d.get('ms_per_step'))"
  done
done

#!/bin/bash
# round 4, GPU call A: the persistent decode step (parity first, then A/B against the launches), configs[3] tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
( timeout 420 python -m pytest tests/test_decode_persistent_gpu.py -x -q -s 2>&1 | tail -40 ) > $O/test_decode_persistent.txt 2>&1
tail -5 $O/test_decode_persistent.txt
if grep -q "passed" $O/test_decode_persistent.txt && ! grep -q "failed\|error" $O/test_decode_persistent.txt; then
  for p in 1 0 1 0; do
    VALLEY_DECODE_PERSISTENT=$p timeout 300 python bench.py --config c5 --decode 256 --warmup 8 --no-cpu-baseline --traffic none --also none 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent=$p', d['value'], d['unit'], d['ms_per_step'], d['roofline']['achieved'])"
  done > $O/decode_ab.txt 2>&1
  cat $O/decode_ab.txt
  VALLEY_DECODE_PERSISTENT=1 timeout 300 python bench.py --config c2 --decode 256 --warmup 8 --no-cpu-baseline --traffic none --also none 2>/dev/null | grep '^{' > $O/decode_7b_persistent.json
  VALLEY_DECODE_PERSISTENT=0 timeout 300 python bench.py --config c2 --decode 256 --warmup 8 --no-cpu-baseline --traffic none --also none 2>/dev/null | grep '^{' > $O/decode_7b_launches.json
  python -c "
import json
for f in ('persistent','launches'):
    d=json.load(open('$O/decode_7b_%s.json'%f)); print('7B', f, d['value'], d['ms_per_step'])
"
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_decode -- python $GRAFT_REPO_ROOT/bench.py --config c5 --decode 64 --warmup 4 --no-cpu-baseline --traffic none --also none > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT; find $O/prof_decode -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
fi
( timeout 600 python -m pytest tests/test_scale_gpu.py -x -q -s -k "c4" 2>&1 | tail -30 ) > $O/test_c4.txt 2>&1
tail -15 $O/test_c4.txt

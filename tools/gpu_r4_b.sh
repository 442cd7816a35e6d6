#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4b; mkdir -p $O
( timeout 420 python -m pytest tests/test_decode_persistent_gpu.py -x -q 2>&1 | tail -6 ) > $O/test_decode_persistent.txt 2>&1
tail -3 $O/test_decode_persistent.txt
timeout 250 python tools/decode_phase_times.py 13b 8 2>&1 | grep -v amdgpu.ids | tee $O/phase_times_13b.txt
VALLEY_HIP_LIB=$PWD/valley_amd/lib/variants/libvalley_hip_ns4.so timeout 250 python tools/decode_phase_times.py 13b 8 2>&1 | grep -v amdgpu.ids | tee $O/phase_times_13b_ns4.txt
for p in 1 0; do
  VALLEY_DECODE_PERSISTENT=$p timeout 300 python bench.py --config c5 --decode 256 --warmup 8 --no-cpu-baseline --traffic none --also none 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent=$p', d['value'], d['unit'], d['ms_per_step'], d['roofline']['achieved'])"
done | tee $O/decode_ab.txt
VALLEY_HIP_LIB=$PWD/valley_amd/lib/variants/libvalley_hip_ns4.so timeout 300 python bench.py --config c5 --decode 256 --warmup 8 --no-cpu-baseline --traffic none --also none 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persistent ns4', d['value'], d['unit'], d['ms_per_step'])"

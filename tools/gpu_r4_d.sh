#!/bin/bash
# round 4, GPU call D: the whole GPU suite on the current build, then the driver-style default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4d; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/gpu_suite.txt 2>&1
tail -6 $O/gpu_suite.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4d/bench_default.json') if l.startswith('{')][-1])
print('c3', d['value'], d['ms_per_step'], d['stages'])
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k,v in d.get('also',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('stages',{}).get('prefill_frac_of_bf16_peak'), v.get('error'))
PY

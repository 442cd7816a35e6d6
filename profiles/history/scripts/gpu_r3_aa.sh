#!/bin/bash
# round-3 GPU call AA: ViT with the residual adds in the out-proj / fc2 epilogues (VALLEY_VIT_RESID_EPI=1) vs add+LN kernels
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/aa3
mkdir -p $O
for F in 128 256 32; do
for E in 0 1 0 1; do
VALLEY_TUNE_CACHE=$O/tune_re.json VALLEY_VIT_RESID_EPI=$E timeout 600 python tools/vit_time.py $F >> $O/vit_re.jsonl 2>> $O/err.txt
done
done
cat $O/vit_re.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['vit_frames'], j['env'].get('VALLEY_VIT_RESID_EPI'), j['ms'], j['frac_bf16_peak'], j['tune_passes'])"
python - <<'PY'
import json
for e in json.load(open("gpurun_out/aa3/tune_re.json")):
    if e["key"][4] == "torch.float32" and e["key"][6]: print(json.dumps(e))
PY
tail -3 $O/err.txt

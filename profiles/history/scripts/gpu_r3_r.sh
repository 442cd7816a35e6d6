#!/bin/bash
# round-3 GPU call R: kernel stats of the 13B decode step (c5) with the norms folded into the GEMVs
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec -o dec -- python $R/bench.py --config c5 --decode 128 --warmup 4 --also none > $O/dec_prof.json 2> $O/prof.err
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3/prof_dec/**/dec_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY

#!/bin/bash
# round-3 GPU call Y: the default bench line again (kernel names as rocprofv3 prints them -> live traffic) + its rocprofv3 stats
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
O=gpurun_out/final
mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_err.txt
head -c 600 $O/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic none --also none > $R/$O/bench_c3_prof.json 2> $R/$O/prof.err
cd $R
find $O/prof_c3 -name "*kernel_trace.csv" -delete; find $O/prof_c3 -name "*.db" -delete

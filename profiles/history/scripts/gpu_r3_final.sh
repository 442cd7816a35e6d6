#!/bin/bash
# round-3 final GPU call: full suite, default bench (live traffic + cpu baseline + also), rocprofv3 stats, PMC of the dominant kernel
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
O=gpurun_out/final
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider 2>&1 | grep -v "^$" > $O/pytest.log
tail -4 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_err.txt
head -c 1500 $O/bench_default.json; echo; tail -3 $O/bench_err.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic none --also none > $R/$O/bench_c3_prof.json 2> $R/$O/prof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_dec -o dec -- python $R/bench.py --config c5 --decode 128 --warmup 4 --also none > $R/$O/bench_dec_prof.json 2> $R/$O/prof_dec.err
cd $R
find $O/prof_dec -name "*kernel_trace.csv" -delete; find $O/prof_dec -name "*.db" -delete
find $O/prof_c3 -name "*kernel_trace.csv" -delete; find $O/prof_c3 -name "*.db" -delete
bash tools/pmc_gemm.sh final_swiglu 2688 27648 5120 tile 197 2 > /dev/null 2>&1; cp gpurun_out/pmc_final_swiglu/summary.txt $O/pmc_swiglu_t197.txt
bash tools/pmc_gemm.sh final_fc1 32768 4096 1024 tile 197 1 > /dev/null 2>&1; cp gpurun_out/pmc_final_fc1/summary.txt $O/pmc_vit_fc1_t197.txt
cat $O/pmc_swiglu_t197.txt $O/pmc_vit_fc1_t197.txt
rm -rf gpurun_out/pmc_final_swiglu gpurun_out/pmc_final_fc1

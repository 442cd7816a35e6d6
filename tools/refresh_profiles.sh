set -x
# REFRESH_RETUNE=1: decide every GEMM shape from scratch first (ignores the shipped table); default: shipped table
if [ "${REFRESH_RETUNE:-0}" = 1 ]; then
export VALLEY_TUNE_TABLE=0 VALLEY_TUNE_TRIALS=5 VALLEY_TUNE_CACHE=$PWD/gpurun_out/tune_final.json
rm -f $VALLEY_TUNE_CACHE
python bench.py --no-cpu-baseline > gpurun_out/f_tune_c2.json 2> gpurun_out/f.err
python bench.py --config c3 --no-cpu-baseline > gpurun_out/f_tune_c3.json 2>> gpurun_out/f.err
python bench.py --config tiny --no-cpu-baseline > gpurun_out/f_tune_tiny.json 2>> gpurun_out/f.err
export VALLEY_TUNE_TRIALS=3
else
: > gpurun_out/f.err
export VALLEY_TUNE_CACHE=$PWD/gpurun_out/tune_final.json      # shipped table + whatever had to be decided online
rm -f $VALLEY_TUNE_CACHE
fi
python bench.py > gpurun_out/f_bench_c2.json 2>> gpurun_out/f.err
python bench.py --config c3 --no-cpu-baseline > gpurun_out/f_bench_c3.json 2>> gpurun_out/f.err
python bench.py --config c3 --decode 256 > gpurun_out/f_dec13.json 2>> gpurun_out/f.err
python bench.py --config c2 --decode 256 > gpurun_out/f_dec7.json 2>> gpurun_out/f.err
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f_prof_c2 -o c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/f_bench_c2_prof.json 2>> $R/gpurun_out/f.err
cd $R
bash tools/pmc_traffic.sh > gpurun_out/f_traffic.txt 2>&1
tail -5 gpurun_out/f.err
python -c "
import json
for f in ['f_bench_c2','f_bench_c3','f_bench_c2_prof']:
    j=json.load(open('gpurun_out/%s.json'%f)); print(f, j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j['config'].get('tune_passes'))
for f in ['f_dec13','f_dec7']:
    j=json.load(open('gpurun_out/%s.json'%f)); print(f, j['value'], j['roofline']['frac'])
"

#!/bin/bash
# round-2 GPU call D: suite, ViT chunk sweep, c4/c2/decode benches, PMC of the 32x32x16 variant
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x 2>&1 | grep -v "^$" | tail -15 > gpurun_out/d_pytest.log
tail -3 gpurun_out/d_pytest.log
python tools/time_pool_variants.py > gpurun_out/d_pool_timing.txt 2>&1; cat gpurun_out/d_pool_timing.txt
B="python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3"
export VALLEY_TUNE_CACHE=$R/gpurun_out/d_tune.json
timeout 900 $B > gpurun_out/d_bench_c3.json 2> gpurun_out/d_err1.txt
VALLEY_VIT_CHUNK=32 timeout 900 $B > gpurun_out/d_bench_c3_chunk32.json 2> gpurun_out/d_err2.txt
VALLEY_VIT_CHUNK=64 timeout 900 $B > gpurun_out/d_bench_c3_chunk64.json 2> gpurun_out/d_err3.txt
timeout 900 $B --config c4 > gpurun_out/d_bench_c4_n1.json 2> gpurun_out/d_err4.txt
timeout 900 $B --config c2 > gpurun_out/d_bench_c2.json 2> gpurun_out/d_err5.txt
timeout 900 python bench.py --config c3 --decode 256 > gpurun_out/d_decode13b.json 2> gpurun_out/d_err6.txt
python - <<'PY'
import json
for f in ("c3", "c3_chunk32", "c3_chunk64", "c4_n1", "c2"):
    try:
        j = json.load(open(f"gpurun_out/d_bench_{f}.json"))
        st = j["stages"]
        print(f, j["value"], "ms", j["ms_per_step"], "vit", st["vit_ms"], st["vit_frames_per_s_per_gpu"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"],
              "tune", j["config"]["tune_passes"], {k: v["TFLOPs"] for k, v in list(j["roofline"]["gemm_shapes"].items())[:9]})
    except Exception as e:
        print(f, "FAILED", e)
j = json.load(open("gpurun_out/d_decode13b.json")); print("decode13b", j["value"], j["roofline"]["frac"])
PY
unset VALLEY_TUNE_CACHE
bash tools/pmc_gemm.sh d_base_c2swiglu 1312 22016 4096 tile 8 2 > /dev/null 2>&1
VALLEY_HIP_LIB=$R/valley_amd/lib/variants/libvalley_hip_m32b.so bash tools/pmc_gemm.sh d_m32b_c2swiglu 1312 22016 4096 tile 8 2 > /dev/null 2>&1
echo "== base"; cat gpurun_out/pmc_d_base_c2swiglu/summary.txt; echo "== m32b"; cat gpurun_out/pmc_d_m32b_c2swiglu/summary.txt
find gpurun_out/pmc_d_* -name "*.csv" -delete; find gpurun_out/pmc_d_* -name "*.db" -delete

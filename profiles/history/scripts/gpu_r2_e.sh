#!/bin/bash
# round-2 GPU call E: suite; same-box A/B of row split and fused RoPE on c3; final default bench; rocprofv3 stats; c2
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x 2>&1 | grep -v "^$" | tail -25 > gpurun_out/e_pytest.log
tail -4 gpurun_out/e_pytest.log
B="python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3"
VALLEY_ROW_SPLIT=0 VALLEY_FUSE_ROPE=0 timeout 900 $B > gpurun_out/e_c3_split0_rope0.json 2> gpurun_out/e_err1.txt
VALLEY_ROW_SPLIT=1 VALLEY_FUSE_ROPE=0 timeout 900 $B > gpurun_out/e_c3_split1_rope0.json 2> gpurun_out/e_err2.txt
VALLEY_TUNE_CACHE=$R/gpurun_out/e_tune.json timeout 900 $B > gpurun_out/e_c3_split1_rope1.json 2> gpurun_out/e_err3.txt
VALLEY_ROW_SPLIT=0 VALLEY_FUSE_ROPE=0 timeout 900 $B > gpurun_out/e_c3_split0_rope0_again.json 2> gpurun_out/e_err4.txt
export VALLEY_TUNE_CACHE=$R/gpurun_out/e_tune.json
timeout 900 $B --config c2 > gpurun_out/e_c2.json 2> gpurun_out/e_err5.txt
python - <<'PY'
import json
for f in ("c3_split0_rope0", "c3_split1_rope0", "c3_split1_rope1", "c3_split0_rope0_again", "c2"):
    try:
        j = json.load(open(f"gpurun_out/e_{f}.json"))
        st = j["stages"]
        print(f, j["value"], "ms", j["ms_per_step"], "vit", st["vit_ms"], st["vit_frames_per_s_per_gpu"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"],
              "tune", j["config"]["tune_passes"], {k: (v["TFLOPs"], v["avg_us"]) for k, v in list(j["roofline"]["gemm_shapes"].items())[:13]})
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 1200 python bench.py > gpurun_out/e_bench_c3_default.json 2> gpurun_out/e_err6.txt
head -c 1500 gpurun_out/e_bench_c3_default.json; echo; tail -c 900 gpurun_out/e_bench_c3_default.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/e_prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic none > $R/gpurun_out/e_bench_c3_prof.json 2> $R/gpurun_out/e_prof.err
cd $R
find gpurun_out/e_prof_c3 -name "*kernel_trace.csv" -delete; find gpurun_out/e_prof_c3 -name "*.db" -delete
f=$(find gpurun_out/e_prof_c3 -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-200

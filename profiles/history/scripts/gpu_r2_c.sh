#!/bin/bash
# round-2 GPU call C: full suite on the new library; A/B of tile order / row split on the c3 bench; final default bench
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rP -x 2>&1 | grep -v "^$" > gpurun_out/c_pytest_full.log
tail -4 gpurun_out/c_pytest_full.log
grep -n "rel-L2\|max-abs\|N4 \|FAILED\|Error" gpurun_out/c_pytest_full.log | head -60
python tools/time_pool_variants.py > gpurun_out/c_pool_timing.txt 2>&1; cat gpurun_out/c_pool_timing.txt
B="python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3"
VALLEY_ROW_SPLIT=0 VLY_TILE_GROUPED=0 timeout 600 $B > gpurun_out/c_bench_split0_grp0.json 2> gpurun_out/c_err1.txt
VALLEY_ROW_SPLIT=0 VLY_TILE_GROUPED=1 timeout 600 $B > gpurun_out/c_bench_split0_grp1.json 2> gpurun_out/c_err2.txt
VALLEY_TUNE_CACHE=$PWD/gpurun_out/c_tune.json timeout 900 $B > gpurun_out/c_bench_split1_grp1.json 2> gpurun_out/c_err3.txt
python - <<'PY'
import json
for f in ("split0_grp0", "split0_grp1", "split1_grp1"):
    try:
        j = json.load(open(f"gpurun_out/c_bench_{f}.json"))
        st = j["stages"]
        print(f, j["value"], "ms", j["ms_per_step"], "vit", st["vit_ms"], st["vit_frac_of_bf16_peak"], "prefill", st["prefill_ms"], st["prefill_frac_of_bf16_peak"],
              "tune", j["config"]["tune_passes"], {k: v["TFLOPs"] for k, v in j["roofline"]["gemm_shapes"].items()})
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/c_err3.txt

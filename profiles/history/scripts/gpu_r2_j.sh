#!/bin/bash
# tile-order A/B on one box: VLY_TILE_GROUPED=0 (round-1 order) vs the default (near-square groups), c3 and c2
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --no-cpu-baseline --traffic none --steps 10 --warmup 3"
VLY_TILE_GROUPED=0 timeout 900 $B > gpurun_out/j_c3_grp0.json 2>/dev/null
timeout 900 $B > gpurun_out/j_c3_grp1.json 2>/dev/null
timeout 900 $B --config c2 > gpurun_out/j_c2_grp1.json 2>/dev/null
timeout 900 $B --config c4 > gpurun_out/j_c4_grp1.json 2>/dev/null
VLY_TILE_GROUPED=0 timeout 900 $B --config c4 > gpurun_out/j_c4_grp0.json 2>/dev/null
python - <<'PY'
import json
for f in ("c3_grp0", "c3_grp1", "c2_grp1", "c4_grp1", "c4_grp0"):
    try:
        j = json.load(open(f"gpurun_out/j_{f}.json"))
        st = j["stages"]
        print(f, j["value"], "ms", j["ms_per_step"], "vit", st.get("vit_ms"), st.get("vit_frames_per_s_per_gpu"), "prefill", st.get("prefill_ms"), st.get("prefill_frac_of_bf16_peak"),
              {k: v["TFLOPs"] for k, v in list(j["roofline"]["gemm_shapes"].items())[:8]})
    except Exception as e:
        print(f, "FAILED", e)
PY
python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" 2>&1 | tail -2

#!/bin/bash
# PMC counters of vly_vit_attention (tools/vit_attn_time.py F), one rocprofv3 pass per counter group (kernel-trace only).
# Usage (GPU box, repo root):  bash tools/pmc_vit_attn.sh [F=128]   -> gpurun_out/pmc_vit_attn/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
F=${1:-128}
out=$R/gpurun_out/pmc_vit_attn
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o p$i -- python $R/tools/vit_attn_time.py $F > /dev/null 2> $out/p$i.err || echo "pass $i failed" >> $out/summary.txt
done
cd $R && python - $out $F <<'PY' >> $out/summary.txt
import csv, glob, sys, collections
agg = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob(sys.argv[1] + "/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "vit_attn" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(sys.argv[1] + "/p1_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "vit_attn" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("vly_vit_attention,", sys.argv[2], "frames; launches", len(dur), "median_us", sorted(dur)[len(dur) // 2] / 1e3 if dur else None)
for k, v in agg.items():
    v = sorted(v)
    print(f"{k:32s} median {v[len(v) // 2]:16.1f}  n={len(v)}")
PY
cat $out/summary.txt

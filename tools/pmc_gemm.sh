#!/bin/bash
# PMC counters of ONE GEMM configuration (tools/gemm_one.py), one rocprofv3 pass per counter group
# (kernel-trace only, never combined with other trace domains).
# Usage (GPU box, repo root):  bash tools/pmc_gemm.sh <tag> M N K tile|sk <hint> [epi]   -> gpurun_out/pmc_<tag>/summary.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" \
           "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o p$i -- python $R/tools/gemm_one.py "$@" > /dev/null 2> $out/p$i.err || echo "pass $i failed" >> $out/summary.txt
done
cd $R && python - $out <<'PY' >> $out/summary.txt
import csv, glob, sys, collections
agg = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob(sys.argv[1] + "/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] or "Cijk" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(sys.argv[1] + "/p1_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] or "Cijk" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if dur:
    print("launches", len(dur), "median_us", sorted(dur)[len(dur) // 2] / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(f"{k:32s} median {v[len(v) // 2]:16.1f}  n={len(v)}")
PY
cat $out/summary.txt

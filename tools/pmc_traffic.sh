#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters, per MI355X_MICROARCH.md §HBM:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slot budget), kernel-trace only.
# Usage (on the GPU box, from the repo root):  bash tools/pmc_traffic.sh [bench args]   -> gpurun_out/pmc_traffic/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic -o $c -- \
      python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events "$@" > /dev/null 2>&1
done
cd $R && python tools/pmc_traffic_summary.py gpurun_out/pmc_traffic

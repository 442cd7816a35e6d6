#!/bin/bash
# Round 5: A/B of the persistent GEMM's tile-boundary variants (tools/ab_lib.py builds under valley_amd/lib/variants/).
#   usage: gpurun -- 'bash tools/r05_boundary_ab.sh "base,ew,zf,ewzf" [diag-names]'   -> gpurun_out/r05_boundary_ab.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
NAMES=${1:-base,ew,zf,ewzf}
DIAG=${2:-}
SHAPES=${SHAPES:-"32768,4096,1024,1,197 32896,3072,1024,0,197|198 32768,1024,1024,0,197 32768,1024,4096,0,197 2688,27648,5120,2,197 2688,15360,5120,0,198 2688,5120,13824,0,198 2688,5120,5120,0,198"}
{
  echo "# checked arms: $NAMES"
  timeout 600 python tools/ab_lib.py run $NAMES $SHAPES
  if [ -n "$DIAG" ]; then
    echo "# diagnostic arms (results not checked): $DIAG"
    AB_NOCHECK=1 timeout 600 python tools/ab_lib.py run $DIAG $SHAPES
  fi
} 2>&1 | tee gpurun_out/r05_boundary_ab.txt

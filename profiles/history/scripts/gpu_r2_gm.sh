#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for shape in "32768 4096 1024 1 9" "32896 3072 1024 0 9" "32768 1024 4096 0 1" "32768 1024 1024 0 9" "2688 27648 5120 2 9" "2688 15360 5120 0 96"; do
  for gm in auto 1 2 4 6 8 16; do
    if [ $gm = auto ]; then python tools/gemm_time.py $shape; else VLY_TILE_GM=$gm python tools/gemm_time.py $shape; fi
  done
done 2>/dev/null | grep shape

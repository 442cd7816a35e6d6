#!/bin/bash
# register-swap epilogue of the 4-wave tiles: tests, then the same launches with VLY_EPILOGUE=lds (own process per setting)
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gemm" 2>&1 | tail -4
for shape in "32768 4096 1024 1 97" "32768 1024 4096 0 97" "32896 3072 1024 0 97" "2688 27648 5120 2 97" "2688 15360 5120 0 98" "2688 5120 13824 0 98" "32768 4096 1024 1 9"; do
  python tools/gemm_time.py $shape 2>/dev/null | grep shape
  VLY_EPILOGUE=lds python tools/gemm_time.py $shape 2>/dev/null | grep shape | sed 's/^/   lds: /'
done

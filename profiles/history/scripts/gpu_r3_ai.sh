#!/bin/bash
# round-3 GPU call AI: last sanity of the rebuilt tree — decode / GEMV / attention unit tests and smoke()
cd ${GRAFT_REPO_ROOT:-.}
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemv or decode_attention or p4_streamk" -p no:cacheprovider 2>&1 | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

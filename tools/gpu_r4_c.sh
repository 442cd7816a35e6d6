#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4c; mkdir -p $O
timeout 250 python tools/debug_decode_persistent4.py 4000 2>&1 | grep -v amdgpu.ids | tail -12 > $O/phase3.txt
cat $O/phase3.txt

#!/bin/bash
# A/B builds of the persistent decode kernel: libvalley_hip.so with decode_step.hip recompiled under extra -D flags
# usage: tools/build_dl_variants.sh name "-DFLAG=1 ..."   -> valley_amd/lib/variants/libvalley_hip_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p valley_amd/lib/variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $@ -c valley_amd/csrc/decode_step.hip -o valley_amd/lib/variants/decode_step_$name.o
objs=$(ls valley_amd/lib/*.o | grep -v decode_step.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o valley_amd/lib/variants/libvalley_hip_$name.so $objs valley_amd/lib/variants/decode_step_$name.o
echo built valley_amd/lib/variants/libvalley_hip_$name.so

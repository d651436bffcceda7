#!/bin/bash
# One differently-compiled kernel TU linked with the standard objects into lib/libblock_aligner_hip_<name>.so (same-box A/B):
#   tools/dev/variant.sh <name> "<extra hipcc flags>" [kind=1] [class=8]
cd "$(dirname "$0")/../../block_aligner_amd/csrc"
name=$1; flags=$2; k=${3:-1}; p=${4:-8}
mkdir -p _build_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp $flags -DBA_KIND=$k -DBA_PMAX=$p -c ba_kernels.hip -o _build_$name/ba_kernels_k${k}_p${p}.o || exit 1
objs=$(ls _build/*.o | grep -v "ba_kernels_k${k}_p${p}.o" | grep -v "ba_host.o")   # (with ba_host_dev.o: the development switches)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libblock_aligner_hip_$name.so $objs _build_$name/ba_kernels_k${k}_p${p}.o
ls -la ../lib/libblock_aligner_hip_$name.so

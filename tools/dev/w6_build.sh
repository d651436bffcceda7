#!/bin/bash
# round 6: k_multi at three waves per SIMD (168 registers) -- kernel unit AND host side built with the same geometry: W waves per workgroup
# (W6_WAVES, default 4: one wave per SIMD and three workgroups per CU. Six-wave workgroups hang the queue even under a six-wave host -- two of them
# do not fit a CU the way the dispatcher spreads their waves, and the persistent kernel waits for workgroups that never become resident).
# W6_EU: waves per SIMD (3; 2 = 256 registers). lib/libblock_aligner_hip_<W6_NAME>.so, development host, DNA class 8 only.
cd "$(dirname "$0")/../../block_aligner_amd/csrc"
F="-DBA_WAVES_PER_WG=${W6_WAVES:-4} -DMQ_WAVES_EU=${W6_EU:-3} $W6_EXTRA"
N=${W6_NAME:-w6}
CXX="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp"
mkdir -p _build_$N
$CXX $F -DBA_KIND=1 -DBA_PMAX=8 -c ba_kernels.hip -o _build_$N/ba_kernels_k1_p8.o || exit 1
$CXX $F -DBA_DEV -DBA_BUILD_ID=\"$(python3 ../../tools/kernel_hash.py)\" -c ba_host.cpp -o _build_$N/ba_host_dev.o || exit 1
objs=$(ls _build/*.o | grep -v "ba_kernels_k1_p8.o" | grep -v "ba_host.o" | grep -v "ba_host_dev.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libblock_aligner_hip_$N.so $objs _build_$N/ba_kernels_k1_p8.o _build_$N/ba_host_dev.o
ls -la ../lib/libblock_aligner_hip_$N.so

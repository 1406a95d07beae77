#!/bin/bash
# A/B build of librtcore with extra compiler flags, beside the shipped one.  usage: tools/build_variant.sh <name> [flags...]  ->  build/librtcore_<name>.so
# (run a tool against it with RTCORE_LIB=$PWD/build/librtcore_<name>.so)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
D=build/$NAME
mkdir -p $D
C="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I raytracing_amd/csrc -Wall -Wno-unused-function $*"
S=raytracing_amd/csrc
$C -c $S/rtcore.hip -o $D/rtcore.o &
$C -ffp-contract=off -c $S/rt_kernels_fp64.hip -o $D/rt_kernels_fp64.o -Rpass-analysis=kernel-resource-usage 2> $D/usage64.txt &
$C -fno-hip-fp32-correctly-rounded-divide-sqrt -c $S/rt_kernels_fp32.hip -o $D/rt_kernels_fp32.o &
$C -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp -c $S/rt_kernels_fp64_obj.hip -o $D/rt_kernels_fp64_obj.o -Rpass-analysis=kernel-resource-usage 2> $D/usage64_obj.txt &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/librtcore_$NAME.so $D/rtcore.o $D/rt_kernels_fp64.o $D/rt_kernels_fp64_obj.o $D/rt_kernels_fp32.o
python3 tools/kernel_usage.py $D/usage64_obj.txt "render_wave_kernel<double, false, true, 3>"

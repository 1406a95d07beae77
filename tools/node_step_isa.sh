#!/bin/bash
# ISA of one wide node visit (rt_wave.h node_step, kQuant): compiles the fp32 kernels to assembly with markers around the visit and
# prints the VALU / memory instruction counts of the OBJ-loader render kernel's copy.   usage: tools/node_step_isa.sh [out.s]
set -e
cd "$(dirname "$0")/../raytracing_amd/csrc"
OUT=${1:-/tmp/rt_kernels_fp32.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I ../../include -I . -Wno-unused-function -fno-hip-fp32-correctly-rounded-divide-sqrt \
    '-DRT_ASM_MARK(text)=asm volatile(text)' --cuda-device-only -S rt_kernels_fp32.hip -o "$OUT" 2>/dev/null
# the second marked region of the file is render_wave_kernel<float, false, true, 3> (the first is the counting twin)
awk '/wide node_step begin/{c++} c==2{print} /wide node_step end/{if(c==2) exit}' "$OUT" > "$OUT.visit"
echo "VALU $(grep -c '^\sv_' "$OUT.visit")  of which v_pk_fma_f32 $(grep -c 'v_pk_fma_f32' "$OUT.visit"), v_cvt_f32_ubyte $(grep -c 'v_cvt_f32_ubyte' "$OUT.visit"), v_cndmask $(grep -c 'v_cndmask' "$OUT.visit")"
echo "global loads $(grep -c 'global_load' "$OUT.visit")  LDS $(grep -c '^\sds_' "$OUT.visit")  SALU $(grep -c '^\ss_' "$OUT.visit")"

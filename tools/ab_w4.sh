# 4 waves per SIMD for the fp64 OBJ variant against the shipped 3, with the stack budgets each needs.  usage (GPU box): bash tools/ab_w4.sh
set -e
export TMPDIR=/tmp
W="--width 3840 --height 2160 --spp 32 --count --mode fp64"
run() { echo "== $1 RT_WIDE_STACK=$2"; RTCORE_LIB=$3 RT_WIDE_STACK=$2 timeout -k 10 200 python3 tools/sweep_wave.py obj $W 2>&1 | grep "^obj"; }
run shipped 35 $PWD/raytracing_amd/librtcore.so
run shipped 31 $PWD/raytracing_amd/librtcore.so
run w4 31 $PWD/build/librtcore_w4.so
run w4 29 $PWD/build/librtcore_w4.so

set -e
export TMPDIR=/tmp
W="--width 3840 --height 2160 --spp 32 --count"
for lib in raytracing_amd/librtcore.so build/librtcore_pair.so; do
  for m in fp64 fp32; do echo "== $lib $m"; RTCORE_LIB=$PWD/$lib timeout -k 10 200 python3 tools/sweep_wave.py obj $W --mode $m 2>&1 | grep "^obj"; done
done

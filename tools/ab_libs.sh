#!/bin/bash
# A/B of alternative builds of librtcore (RTCORE_LIB) on the mesh and sphere workloads at 4K.
# usage: tools/ab_libs.sh <mode fp32|fp64> lib1.so lib2.so ...   (paths relative to the repo root)
set -e
MODE=$1; shift
for lib in "$@"; do
  echo "== $lib"
  RTCORE_LIB=$PWD/$lib timeout -k 10 250 python3 tools/sweep_wave.py obj --width 3840 --height 2160 --spp 8 --count --mode $MODE 2>&1 | grep "^obj"
  RTCORE_LIB=$PWD/$lib timeout -k 10 150 python3 tools/sweep_wave.py spheres --width 3840 --height 2160 --spp 32 --count --mode $MODE 2>&1 | grep "^spheres"
done

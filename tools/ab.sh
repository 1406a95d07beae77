#!/bin/bash
# A/B of the fast-mode kernels on the three workloads at 4K (prints kernel ms; lower is better).  usage: tools/ab.sh [extra sweep args]
set -e
W="--width 3840 --height 2160"
timeout -k 10 150 python3 tools/sweep_wave.py spheres $W --spp 32 --count "$@"
timeout -k 10 250 python3 tools/sweep_wave.py obj $W --spp 8 --count "$@"
timeout -k 10 150 python3 tools/sweep_wave.py deep $W --spp 16 --depth 200 --count "$@"

# A/B of the decoupled kernel (RT_FLAG_QUEUE = 64) against the default wave kernel at 4K.  usage: bash tools/queue_ab.sh [spp]
W="--width 3840 --height 2160"
SPP=${1:-64}
for mode in fp32 fp64; do
  echo "== obj $mode default"; timeout -k 10 300 python3 tools/sweep_wave.py obj $W --spp $SPP --count --mode $mode 2>&1 | grep "^obj"
  echo "== obj $mode queue"; RT_BVH_DEPTH=24 timeout -k 10 300 python3 tools/sweep_wave.py obj $W --spp $SPP --count --mode $mode --flags 64 2>&1 | grep "^obj"
done

W="--width 3840 --height 2160"
for nk in 128 176 224; do for ho in 8 16 32; do
  echo "== node_keep $nk handoff $ho"; RT_Q_HANDOFF=$ho RT_BVH_DEPTH=24 timeout -k 10 300 python3 tools/sweep_wave.py obj $W --spp 32 --count --mode fp32 --flags 64 --node $nk 2>&1 | grep "^obj"
done; done
for sl in 128 256; do echo "== slots $sl"; RT_Q_SLOTS=$sl RT_BVH_DEPTH=24 timeout -k 10 300 python3 tools/sweep_wave.py obj $W --spp 32 --count --mode fp32 --flags 64 2>&1 | grep "^obj"; done

set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
for leaf in 4 2 1; do for cp in 1.0 2.0 3.0; do
  echo "== leaf $leaf cprim $cp" >> $O/sweep_tree.log
  RT_BVH_LEAF=$leaf RT_BVH_CPRIM=$cp timeout -k 10 120 python3 tools/sweep_wave.py obj --mode fp64 --spp 32 --width 3840 --height 2160 --count 2>&1 | grep "^obj" >> $O/sweep_tree.log
done; done
cat $O/sweep_tree.log

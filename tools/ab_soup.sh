# the 10 M-triangle soup (BASELINE configs[3]) under tree / visit variants.  usage (GPU box): bash tools/ab_soup.sh <tag>
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
(while true; do sleep 45; date >> $O/heartbeat.txt; done) & HB=$!
A="--workload soup --spp 8 --steps 1 --warmup 1 --no-cpu --no-fp32 --image-width 0"
run() { name=$1; shift; env "$@" timeout -k 10 700 python3 $GRAFT_REPO_ROOT/bench.py $A > $O/soup_$name.json 2> $O/soup_$name.err; echo "$name rc=$?" >> $O/ab_soup.txt; }
run shipped RT_NOP=1
run leaf4 RT_BVH_LEAF=4
run sorted RTCORE_LIB=$GRAFT_REPO_ROOT/build/librtcore_sorted.so
kill $HB
cat $O/ab_soup.txt

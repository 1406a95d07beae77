# bench.py on the 10 M-triangle soup (BASELINE configs[3]) at its stated size; a heartbeat file keeps the run alive through the host-side build.
# usage (GPU box): bash tools/bench_soup.sh <tag> [spp, default 16]   -> gpurun_out/<tag>/bench_soup_<spp>.json
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
SPP=${2:-16}
mkdir -p $O
(while true; do sleep 45; date >> $O/heartbeat.txt; done) & HB=$!
timeout -k 10 1000 python3 $GRAFT_REPO_ROOT/bench.py --workload soup --spp $SPP --steps 2 --warmup 1 --image-width 0 > $O/bench_soup_$SPP.json 2> $O/bench_soup_$SPP.err
RC=$?
kill $HB
echo "rc=$RC"
cut -c1-300 $O/bench_soup_$SPP.json
exit $RC

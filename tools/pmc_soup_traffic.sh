# Memory-side traffic of the render kernel on the 10 M-triangle soup: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes of the bench
# command (bench.py itself after `--`), a heartbeat file through the host-side builds.  usage (GPU box): bash tools/pmc_soup_traffic.sh <tag>
# then here:  python tools/hbm_traffic.py gpurun_out/<tag>/soup_fetch/fetch_results.db gpurun_out/<tag>/soup_write/write_results.db <soup bench line> --round <tag>
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp
(while true; do sleep 45; date >> $O/heartbeat.txt; done) & HB=$!
ARGS="--workload soup --spp 16 --steps 1 --warmup 0 --no-cpu --no-fp32 --image-width 0"
timeout -s KILL 520 rocprofv3 --pmc FETCH_SIZE -d $O/soup_fetch -o fetch -- python3 $B $ARGS > $O/soup_fetch.txt 2>&1; RC1=$?
timeout -s KILL 520 rocprofv3 --pmc WRITE_SIZE -d $O/soup_write -o write -- python3 $B $ARGS > $O/soup_write.txt 2>&1; RC2=$?
kill $HB
echo "fetch rc=$RC1 write rc=$RC2"; ls $O

# Memory latency and cache hit counters of the render kernel (one rocprofv3 --pmc pass each; bench.py itself after `--`).
# usage (on the GPU box): bash tools/pmc_latency.sh <tag> [bench args...]   -> gpurun_out/<tag>/{lat,tcp,tcc}
set -e
export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
B=$GRAFT_REPO_ROOT/bench.py
ARGS="${*:---spp 64 --no-cpu --no-fp32 --image-width 0}"
cd /tmp
timeout -s KILL 170 rocprofv3 --pmc VmemLatency -d $O/lat -o lat -- python3 $B $ARGS > $O/lat.txt 2>&1
timeout -s KILL 170 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum -d $O/tcp -o tcp -- python3 $B $ARGS > $O/tcp.txt 2>&1
timeout -s KILL 170 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d $O/tcc -o tcc -- python3 $B $ARGS > $O/tcc.txt 2>&1
ls $O

# Round-3 measurements on the GPU box.  usage: bash tools/run_r03_measurements.sh <tag> <part: bench|pmc|pmc_sorted|spheres>
# (the 10 M-triangle soup has scripts of its own -- its host-side build outlasts gpurun's silence limit without a heartbeat: tools/bench_soup.sh,
# tools/pmc_soup_traffic.sh; memory latency and cache hit rates of the mesh kernel: tools/pmc_latency.sh; summaries into profiles/: tools/collect_r03.py)
# Every profiled command is `python3 bench.py ...` itself after `--` (no shell or env hop under rocprofv3); PMC passes are separate runs.
set -e
export TMPDIR=/tmp
TAG=${1:-r03}
PART=${2:-bench}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp
P64="--spp 64 --no-cpu --no-fp32 --image-width 0"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
TA="TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"
ST="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"
pmc() {   # pmc <name> <counters> <bench args...>
  local name=$1 ctr=$2; shift 2
  timeout -s KILL 170 rocprofv3 --pmc $ctr -d $O/$name -o $name -- python3 $B "$@" > $O/$name.txt 2>&1
}
case $PART in
bench)
  timeout -k 10 400 python3 $B > $O/bench_obj_1024.json 2> $O/bench_obj_1024.err
  timeout -k 10 200 python3 $B --workload random_spheres --spp 256 > $O/bench_spheres_256.json 2> $O/bench_spheres_256.err
  timeout -k 10 150 python3 $B $P64 > $O/bench_obj_64.json 2> $O/bench_obj_64.err
  timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $B $P64 > $O/trace_stdout.txt 2>&1
  ;;
pmc)
  pmc sq "$SQ" $P64
  pmc ta "$TA" $P64
  pmc st "$ST" $P64
  pmc fetch "FETCH_SIZE" $P64
  pmc write "WRITE_SIZE" $P64
  ;;
pmc_sorted)      # the same passes over the A/B library with round 2's sorted visit (build/librtcore_sorted.so, make EXTRA=-DRT_SORTED_VISIT)
  export RTCORE_LIB=$GRAFT_REPO_ROOT/build/librtcore_sorted.so
  timeout -k 10 150 python3 $B $P64 > $O/bench_obj_64_sorted.json 2> $O/bench_obj_64_sorted.err
  pmc sq_sorted "$SQ" $P64
  pmc ta_sorted "$TA" $P64
  pmc st_sorted "$ST" $P64
  ;;
spheres)
  S64="--workload random_spheres --spp 64 --no-cpu --no-fp32 --image-width 0"
  timeout -k 10 150 python3 $B $S64 > $O/bench_spheres_64.json 2> $O/bench_spheres_64.err
  pmc sph_sq "$SQ" $S64
  pmc sph_ta "$TA" $S64
  pmc sph_st "$ST" $S64
  ;;
esac
ls $O | head -80

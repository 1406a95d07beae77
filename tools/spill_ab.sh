# WRITE_SIZE and kernel time of the OBJ-loader wave kernel with and without register spills.
# usage (GPU box): bash tools/spill_ab.sh <tag>
set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp
run() {  # name lib mode
  export RTCORE_LIB=$GRAFT_REPO_ROOT/raytracing_amd/$2
  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/$1_w -o w -- python3 $B --mode $3 --spp 16 --steps 1 --warmup 0 --no-cpu --no-fp32 > $O/$1_w.txt 2>&1
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/$1_f -o f -- python3 $B --mode $3 --spp 16 --steps 1 --warmup 0 --no-cpu --no-fp32 > $O/$1_f.txt 2>&1
  timeout -k 10 150 python3 $B --mode $3 --spp 16 --steps 2 --warmup 1 --no-cpu --no-fp32 > $O/$1_bench.json 2> $O/$1_bench.err
}
run fp32_5waves_spill librtcore.so fp32
run fp32_4waves_nospill librtcore_o4.so fp32
run fp64_4waves_spill librtcore.so fp64
run fp64_2waves_nospill librtcore_w2.so fp64
ls $O

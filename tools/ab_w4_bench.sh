# the bench line of the 4-waves-per-SIMD fp64 OBJ build (build/librtcore_w4.so, tools/build_variant.sh w4 -DRT_WAVES_FP64_OBJ=4; its stack budget 31 lets
# four workgroups share a CU's LDS) at 64 and 1024 spp, and its PMC occupancy pass.  usage (GPU box): bash tools/ab_w4_bench.sh <tag>
set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
B=$GRAFT_REPO_ROOT/bench.py
export RTCORE_LIB=$GRAFT_REPO_ROOT/build/librtcore_w4.so RT_WIDE_STACK=31
cd /tmp
timeout -k 10 150 python3 $B --spp 64 --no-cpu --no-fp32 --image-width 0 > $O/w4_bench_obj_64.json 2> $O/w4_bench_obj_64.err
timeout -k 10 300 python3 $B --spp 1024 --no-cpu --no-fp32 --image-width 0 > $O/w4_bench_obj_1024.json 2> $O/w4_bench_obj_1024.err
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY -d $O/w4_sq -o w4_sq -- python3 $B --spp 64 --no-cpu --no-fp32 --image-width 0 > $O/w4_sq.txt 2>&1
timeout -s KILL 170 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE -d $O/w4_ta -o w4_ta -- python3 $B --spp 64 --no-cpu --no-fp32 --image-width 0 > $O/w4_ta.txt 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU -d $O/w4_st -o w4_st -- python3 $B --spp 64 --no-cpu --no-fp32 --image-width 0 > $O/w4_st.txt 2>&1
ls $O

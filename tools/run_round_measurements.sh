# Round measurements on the GPU box (one gpurun call): bench lines, kernel trace, PMC passes.  usage: bash tools/run_round_measurements.sh r02a
set -e
export TMPDIR=/tmp
TAG=${1:-r02}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 300 python $B > $O/bench_obj_1024.json 2> $O/bench_obj_1024.err
timeout -k 10 200 python $B --workload random_spheres --spp 256 > $O/bench_spheres_256.json 2> $O/bench_spheres.err
cd /tmp
PROF="--spp 64 --no-cpu --no-fp32"
timeout -k 10 150 python3 $B $PROF > $O/bench_obj_64.json 2> $O/bench_obj_64.err
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $B $PROF > $O/trace_stdout.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python3 $B $PROF > $O/pmc_fetch.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o write -- python3 $B $PROF > $O/pmc_write.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY -d $O/pmc_sq -o sq -- python3 $B --spp 16 --steps 1 --warmup 0 --no-cpu --no-fp32 > $O/pmc_sq.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE -d $O/pmc_ta -o ta -- python3 $B --spp 16 --steps 1 --warmup 0 --no-cpu --no-fp32 > $O/pmc_ta.txt 2>&1
SPH="--workload random_spheres --spp 16 --steps 1 --warmup 0 --no-cpu --no-fp32 --image-width 0"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY -d $O/sph_sq -o sq -- python3 $B $SPH > $O/sph_sq.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE -d $O/sph_ta -o ta -- python3 $B $SPH > $O/sph_ta.txt 2>&1
ls -R $O | head -60

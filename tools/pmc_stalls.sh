# PMC passes on where the waves of the render kernel spend their cycles.  usage (GPU box): bash tools/pmc_stalls.sh <tag> [extra bench.py arguments, e.g. --workload random_spheres]
set -e
export TMPDIR=/tmp
TAG=${1:-r02}
shift || true
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp
ARGS="--spp 16 --steps 1 --warmup 0 --no-cpu --no-fp32 --image-width 0 $*"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH -d $O/st1 -o st1 -- python3 $B $ARGS > $O/st1.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/st2 -o st2 -- python3 $B $ARGS > $O/st2.txt 2>&1
ls $O

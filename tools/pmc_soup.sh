# PMC passes of the render kernel on the 10 M-triangle soup (BASELINE configs[3]).  usage (on the GPU box): bash tools/pmc_soup.sh <tag>
set -e
export TMPDIR=/tmp
TAG=${1:-r02}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp
ARGS="--workload soup --spp 16 --steps 1 --warmup 0 --no-cpu --no-fp32 --image-width 0"
timeout -s KILL 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY -d $O/soup_sq -o sq -- python3 $B $ARGS > $O/soup_sq.txt 2>&1
timeout -s KILL 500 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE -d $O/soup_ta -o ta -- python3 $B $ARGS > $O/soup_ta.txt 2>&1
ls -R $O | head -30

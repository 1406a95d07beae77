set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r01m_pmc
mkdir -p $O
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY -d $O/sq -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --spp 16 --steps 1 --warmup 0 --no-cpu > $O/sq.txt 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE -d $O/ta -o ta -- python3 $GRAFT_REPO_ROOT/bench.py --spp 16 --steps 1 --warmup 0 --no-cpu > $O/ta.txt 2>&1
ls -R $O | head

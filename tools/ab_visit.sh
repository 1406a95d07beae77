# per-tree visiting order (Scene::sorted_visit, chosen by rtcore.hip's census): the mesh must keep its slot-order speed, the 10 M-triangle soup must get the sorted visit's.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
(while true; do sleep 45; date >> $O/heartbeat.txt; done) & HB=$!
timeout -k 10 300 python -u -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k "visiting_orders or quantised or first_hits_are_the_reference" > $O/visit_tests.txt 2>&1; echo "tests rc=$?" >> $O/ab_visit.txt
W="--width 3840 --height 2160 --spp 32 --count"
for m in fp64 fp32; do RT_DEBUG=1 timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/sweep_wave.py obj $W --mode $m 2>&1 | grep -E "^obj|census" >> $O/ab_visit.txt; done
RT_DEBUG=1 timeout -k 10 700 python3 $GRAFT_REPO_ROOT/bench.py --workload soup --spp 8 --steps 1 --warmup 1 --no-cpu --no-fp32 --image-width 0 > $O/soup_auto.json 2> $O/soup_auto.err; echo "soup rc=$?" >> $O/ab_visit.txt
grep census $O/soup_auto.err >> $O/ab_visit.txt
kill $HB
cat $O/ab_visit.txt; tail -3 $O/visit_tests.txt

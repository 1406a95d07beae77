set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r01m
mkdir -p $O
timeout -k 10 500 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $O/pytest_gpu.txt 2>&1 || { tail -20 $O/pytest_gpu.txt; exit 1; }
tail -2 $O/pytest_gpu.txt
timeout -k 10 200 python bench.py > $O/bench_obj_1024.json 2> $O/bench_obj_1024.err
timeout -k 10 100 python bench.py --workload random_spheres --spp 256 --no-cpu > $O/bench_spheres_256.json 2> $O/bench_spheres.err
timeout -k 10 100 python bench.py --spp 64 --no-cpu > $O/bench_obj_64.json 2> $O/bench_obj_64.err
cd /tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --spp 64 --no-cpu > $O/trace_stdout.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --spp 64 --no-cpu > $O/pmc_fetch.txt 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o write -- python3 $GRAFT_REPO_ROOT/bench.py --spp 64 --no-cpu > $O/pmc_write.txt 2>&1
ls -R $O | head -40

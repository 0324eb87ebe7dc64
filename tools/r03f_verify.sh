python -m pytest tests -m gpu -x -q > gpurun_out/r03f_tests.log 2>&1; tail -3 gpurun_out/r03f_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/r03f_bench_default.json 2> gpurun_out/r03f_bench_default.err; tail -c 200 gpurun_out/r03f_bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03f_prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-ops --no-end-to-end --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03f_bench_under_profiler.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03f_prof.err
head -12 $(find $GRAFT_REPO_ROOT/gpurun_out/r03f_prof -name "*kernel_stats.csv" | head -1) | cut -c1-150

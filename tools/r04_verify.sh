#!/bin/bash
# Round-4 evidence at ONE commit on one box: tools/r04_verify.sh TAG  (e.g. r04f) -> gpurun_out/TAG_*; copy what is to be judged
# into profiles/.  Under rocprofv3 the program goes directly after `--` (no env / bash hops).
tag=$1
root=${GRAFT_REPO_ROOT:-$PWD}
o=$root/gpurun_out
cd $root
python bench.py > $o/${tag}_bench_default.json 2> $o/${tag}_bench_default.err; tail -c 300 $o/${tag}_bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $o/${tag}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -o b -- python3 $root/bench.py --steps 5 --warmup 2 --no-ops --no-end-to-end --no-cpu-baseline > $o/${tag}_bench_under_profiler.json 2> $o/${tag}_prof.err
cp $(find $o/${tag}_prof -name "*kernel_stats.csv" | head -1) $o/${tag}_bench_kernel_stats.csv
head -14 $o/${tag}_bench_kernel_stats.csv | cut -c1-150
cd $root
tools/pmc_bench.sh ${tag}_FETCH "FETCH_SIZE" --steps 2 --warmup 1 > /dev/null 2>&1
tools/pmc_bench.sh ${tag}_WRITE "WRITE_SIZE" --steps 2 --warmup 1 > /dev/null 2>&1
python3 tools/pmc_traffic.py $o/pmc_${tag}_FETCH $o/pmc_${tag}_WRITE $o/${tag} 2>&1 | tail -3
tools/pmc_bench.sh ${tag}_SQ "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" --steps 2 --warmup 1 > /dev/null 2>&1
cp $o/pmc_${tag}_SQ/summary.txt $o/${tag}_pmc_sq_bench.txt
head -12 $o/${tag}_pmc_sq_bench.txt | cut -c1-200

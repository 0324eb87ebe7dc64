#!/bin/bash
# Round-3 profile set (run on the GPU box from the repo root): kernel stats of the bench command and of R-L on the deskewed
# volumes, HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) and LDS bank-conflict counters.  Outputs under gpurun_out/.
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r03_prof_bench -o b -- python3 $root/bench.py --steps 5 --warmup 2 --no-ops --no-end-to-end --no-cpu-baseline > $root/gpurun_out/r03_bench_under_profiler.json 2> $root/gpurun_out/r03_prof_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r03_prof_awk -o rl -- python3 $root/tools/rl_awkward_profile.py big > $root/gpurun_out/r03_rl_awkward.log 2>&1
cd $root
tools/pmc_bench.sh r03_FETCH "FETCH_SIZE" --steps 2 --warmup 1 > /dev/null
tools/pmc_bench.sh r03_WRITE "WRITE_SIZE" --steps 2 --warmup 1 > /dev/null
python3 tools/pmc_traffic.py gpurun_out/pmc_r03_FETCH gpurun_out/pmc_r03_WRITE gpurun_out/r03a
tools/pmc_cmd.sh r03_LDS "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" tools/rl_awkward_profile.py big > gpurun_out/r03_pmc_lds_awkward.txt
tools/pmc_bench.sh r03_LDSB "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --steps 2 --warmup 1 > gpurun_out/r03_pmc_lds_bench.txt

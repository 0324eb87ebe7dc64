#!/bin/bash
# rocprofv3 kernel stats of the default bench command: tools/prof_bench.sh NAME [bench args] -> gpurun_out/prof_NAME/
set -e
name=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py --no-cpu-baseline --no-end-to-end --no-ops "$@" > $out/bench.log 2>&1 || (tail -20 $out/bench.log; exit 1)
cd $root
grep '^{' $out/bench.log | tail -1 > $out/bench.json || true
python3 tools/show_stats.py $out 16

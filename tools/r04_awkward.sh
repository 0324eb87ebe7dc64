#!/bin/bash
# tools/r04_awkward.sh TAG [big]: R-L x10 on the deskewed volumes under the kernel trace -> gpurun_out/prof_TAG
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_${tag}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/rl_awkward_profile.py "$@" > $out/run.log 2>&1; rc=$?
cd $root
grep "RL x10" $out/run.log
python3 tools/show_stats.py $out 24
exit $rc

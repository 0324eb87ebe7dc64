#!/bin/bash
# tools/r04_cubic.sh TAG: spline + flat-field tests, flat-field A/B (stock vs variant ffold), the cubic warp under the kernel trace
tag=$1
root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "flat_field or median or spline or cubic" > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
if [ -z "$SKIP_FF" ]; then
  for lib in stock ffold stock ffold; do
    echo "== $lib" >> gpurun_out/${tag}_ff.log
    if [ $lib = stock ]; then unset BHCORE_LIB; else export BHCORE_LIB=$root/biahub_amd/build/variants/libbhcore_$lib.so; fi
    timeout -k 10 200 python3 tools/time_flatfield.py >> gpurun_out/${tag}_ff.log 2>&1; rc=$?
    if [ $rc -ge 124 ]; then exit $rc; fi
  done
  unset BHCORE_LIB
  grep -v amdgpu.ids gpurun_out/${tag}_ff.log
fi
out=$root/gpurun_out/prof_${tag}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/time_cubic.py > $out/run.log 2>&1; rc=$?
cd $root
grep -E "^(cubic|linear)" $out/run.log
python3 tools/show_stats.py $out 12


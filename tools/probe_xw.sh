#!/bin/bash
# per-kernel times of the X passes under the timing probes (wrong results): tools/probe_xw.sh
set -e
for v in default p1 p2 p7; do
  if [ $v = default ]; then unset BHCORE_LIB; else export BHCORE_LIB=$PWD/biahub_amd/build/variants/libbhcore_$v.so; fi
  echo "== $v"
  bash tools/prof_bench.sh probe_$v --steps 2 --warmup 1 2>/dev/null | grep "xw_kernel<10, [45]>" | awk '{print $3, $4, $(NF-3), $(NF-2), $(NF-1)}'
done

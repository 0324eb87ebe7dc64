#!/bin/bash
# one-pass fill (tile / persistent kernel) against the mask pipeline at the bench shape, same box (HIP-event times of the library's timers)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/r04_deskew_ab.txt
: > $out
for r in 1 2; do
  echo "== one pass, tile kernel (default)" >> $out
  python3 $root/tools/time_deskew.py 2>&1 | grep fill= >> $out
  echo "== one pass, persistent kernel (BH_DESKEW_ROWS_KERNEL=pers)" >> $out
  BH_DESKEW_ROWS_KERNEL=pers python3 $root/tools/time_deskew.py 2>&1 | grep fill= >> $out
  echo "== mask pipeline (BH_DESKEW_ONEPASS=0)" >> $out
  BH_DESKEW_ONEPASS=0 python3 $root/tools/time_deskew.py 2>&1 | grep fill= >> $out
done
cat $out

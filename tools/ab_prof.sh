#!/bin/bash
# interleaved same-box A/B of library variants under rocprofv3 (per-kernel times): tools/ab_prof.sh OUT ROUNDS NAME1 NAME2 ...
# ("stock" = the in-tree library).  Process-to-process spread on one box is several percent (physical page placement), so the
# variants alternate ROUNDS times and the summary prints every run.
out=$1; rounds=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
: > $root/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
for r in $(seq 1 $rounds); do
  for name in "$@"; do
    if [ $name = stock ]; then unset BHCORE_LIB; else export BHCORE_LIB=$root/biahub_amd/build/variants/libbhcore_$name.so; fi
    rm -rf /tmp/prof_ab
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-ops --no-end-to-end --no-cpu-baseline > /tmp/prof_ab.json 2> /tmp/prof_ab.err
    f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$name" >> $root/gpurun_out/$out <<'PY'
import csv, sys
rows = {r['Name']: float(r['AverageNs']) / 1e6 for r in csv.DictReader(open(sys.argv[1]))}
def pick(sub):
    for k, v in rows.items():
        if sub in k:
            return v
    return float('nan')
print(f"{sys.argv[2]:10s} colz5 {pick('colz_kernel<5>'):6.3f}  colw {pick('colw_kernel<10, 0>'):6.3f}  xw4 {pick('xw_kernel<10, 4>'):6.3f}  xw5 {pick('xw_kernel<10, 5>'):6.3f}  xw3 {pick('xw_kernel<10, 3>'):6.3f}  deskew {pick('deskew_pers_kernel'):6.3f}  rowsums {pick('row_sums_kernel'):6.3f}")
PY
    tail -n 1 $root/gpurun_out/$out
  done
done

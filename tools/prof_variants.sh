#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of the stock library and of tuning variants on ONE box:
#   tools/prof_variants.sh OUTFILE NAME1 [NAME2 ...]      (variants built by tools/build_variant.py; "stock" = the in-tree library)
# Prints the top kernels of `bench.py --steps 2 --warmup 1 --no-ops --no-end-to-end --no-cpu-baseline` per variant.
out=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out
: > $root/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  if [ $name = stock ]; then unset BHCORE_LIB; else export BHCORE_LIB=$root/biahub_amd/build/variants/libbhcore_$name.so; fi
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-ops --no-end-to-end --no-cpu-baseline > /tmp/prof_$name.json 2> /tmp/prof_$name.err
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  { echo "== $name"; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:64]:64s} {r['Calls']:>4s} {float(r['AverageNs'])/1e6:8.3f} ms")
PY
  } >> $root/gpurun_out/$out 2>&1
  tail -n 9 $root/gpurun_out/$out
done

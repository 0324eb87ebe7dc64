#!/bin/bash
# per-kernel times of the stock library and of a variant under rocprofv3: tools/prof_variant.sh NAME
name=$1
cd /tmp && export TMPDIR=/tmp
for lib in stock $name; do
  if [ $lib != stock ]; then export BHCORE_LIB=$GRAFT_REPO_ROOT/biahub_amd/build/variants/libbhcore_$name.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$lib -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-ops --no-end-to-end --no-cpu-baseline > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>4s} {float(r['AverageNs'])/1e6:8.3f} ms")
PY
done

#!/bin/bash
# per-kernel times under rocprofv3 for several environment settings on one box: tools/prof_env.sh "cfg1" "cfg2" ... ("-" = none)
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1)); c=$cfg; [ "$c" = "-" ] && c=""
  d=$GRAFT_REPO_ROOT/gpurun_out/prof_env_$i; rm -rf $d
  for kv in $c; do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-ops --no-end-to-end --no-cpu-baseline > $d.json 2> $d.err
  for kv in $c; do unset ${kv%%=*}; done
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== ${c:-default}"; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>4s} {float(r['AverageNs'])/1e6:8.3f} ms")
PY
done

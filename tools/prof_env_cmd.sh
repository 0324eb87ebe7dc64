#!/bin/bash
# per-kernel times of ANY python command under rocprofv3 for several environment settings on one box:
#   tools/prof_env_cmd.sh "tools/rl_awkward_profile.py big" "cfg1" "cfg2" ...   ("-" = none)
cmd=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1)); c=$cfg; [ "$c" = "-" ] && c=""
  d=$GRAFT_REPO_ROOT/gpurun_out/prof_cmd_$i; rm -rf $d
  for kv in $c; do export $kv; done
  (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $cmd > $d.log 2> $d.err)
  for kv in $c; do unset ${kv%%=*}; done
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== ${c:-default}"; grep "RL x10" $d.log | tail -2; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>4s} {float(r['AverageNs'])/1e6:8.3f} ms")
PY
done

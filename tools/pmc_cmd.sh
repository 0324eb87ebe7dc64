#!/bin/bash
# One rocprofv3 counter pass over a python script: tools/pmc_cmd.sh NAME "CTR1 CTR2 ..." script.py [args]
# (counters alone with --kernel-trace: no other trace domains).  Per-kernel means -> gpurun_out/pmc_NAME/summary.txt
set -e
name=$1; ctrs=$2; script=$3; shift 3
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmc_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $root/$script "$@" > $out/run.log 2>&1 || (tail -20 $out/run.log; exit 1)
cd $root
python3 - "$out" <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(58), "calls", " ".join(n.rjust(22) for n in names))
for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values())):
    if "bh::" not in k:
        continue
    n = len(next(iter(acc[k].values())))
    print(k[:58].ljust(58), str(n).rjust(5), " ".join(("%.4g" % (sum(acc[k][c]) / max(1, len(acc[k][c])))).rjust(22) for c in names))
PY

set -e
mkdir -p gpurun_out/ab
for v in xw tile; do
  if [ $v = xw ]; then unset BH_FC_XW; else export BH_FC_XW=0; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-ops > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || (tail -5 gpurun_out/ab/$v.err; exit 1)
  python - <<PY
import json
r=json.load(open("gpurun_out/ab/$v.json"))
print("$v", "ms/step %.1f"%r["ms_per_step"], "rl iter ms %.2f"%r["roofline"]["ms"], "frac %.3f"%r["roofline"]["frac"])
PY
done
unset BH_FC_XW
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ab/prof -o xw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-ops > $GRAFT_REPO_ROOT/gpurun_out/ab/prof.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/show_stats.py gpurun_out/ab/prof 14 || find gpurun_out/ab/prof | head

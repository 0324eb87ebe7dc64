#!/bin/bash
# same-box A/B over an environment switch: tools/ab_env.sh VAR "v1 v2 ..." [bench args]
set -e
var=$1; vals=$2; shift 2
mkdir -p gpurun_out/ab
for v in $vals; do
  export $var=$v
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-ops "$@" > gpurun_out/ab/env_$v.json 2> gpurun_out/ab/env_$v.err || (tail -5 gpurun_out/ab/env_$v.err; exit 1)
  python - <<PY
import json
r=json.load(open("gpurun_out/ab/env_$v.json"))
print("$var=$v".ljust(18), "ms/step %.1f"%r["ms_per_step"], "rl iter ms %.2f"%r["roofline"]["ms"])
PY
done

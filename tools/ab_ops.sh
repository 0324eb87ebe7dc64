#!/bin/bash
# same-box A/B of the ops block: tools/ab_ops.sh NAME1 [NAME2 ...] (variants built by tools/build_variant.py)
set -e
mkdir -p gpurun_out/ab
for v in default "$@"; do
  if [ $v = default ]; then unset BHCORE_LIB; else export BHCORE_LIB=$PWD/biahub_amd/build/variants/libbhcore_$v.so; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/ab/ops_$v.json 2> gpurun_out/ab/ops_$v.err || (tail -5 gpurun_out/ab/ops_$v.err; exit 1)
  python - <<PY
import json
r=json.load(open("gpurun_out/ab/ops_$v.json"))
print("$v".ljust(12), "rl %.2f"%r["roofline"]["ms"], " ".join("%s %.2f"%(k[:14], x["ms"]) for k,x in r["ops"].items() if "affine" not in k and "copy" not in k))
PY
done

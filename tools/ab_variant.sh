#!/bin/bash
# same-box A/B of the stock library against tuning variants: tools/ab_variant.sh NAME1 [NAME2 ...] (built by tools/build_variant.py)
set -e
mkdir -p gpurun_out/ab
for v in default "$@" default; do
  if [ $v = default ]; then unset BHCORE_LIB; else export BHCORE_LIB=$PWD/biahub_amd/build/variants/libbhcore_$v.so; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-ops > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || (tail -5 gpurun_out/ab/$v.err; exit 1)
  python - <<PY
import json
r=json.load(open("gpurun_out/ab/$v.json"))
print("$v".ljust(16), "ms/step %.1f"%r["ms_per_step"], "rl iter ms %.2f"%r["roofline"]["ms"], "deskew %.2f fill %.2f"%(r["roofline_deskew"]["ms"], r["roofline_deskew"]["fill_passes_ms"]))
PY
done

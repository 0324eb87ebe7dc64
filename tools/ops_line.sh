#!/bin/bash
# bench line with the `ops` block, printed as a table: tools/ops_line.sh NAME
set -e
mkdir -p gpurun_out/$1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/$1/ops.json 2> gpurun_out/$1/ops.err || (tail -20 gpurun_out/$1/ops.err; exit 1)
python - "$1" <<'PY'
import json, sys
r = json.load(open(f"gpurun_out/{sys.argv[1]}/ops.json"))
for k, v in r["ops"].items():
    print(k.ljust(24), "%8.2f ms %7.0f GB/s %5.1f%%" % (v["ms"], v["GBps"], 100 * v["frac"]))
d = r["roofline_deskew"]
print("deskew kernel %.2f ms (%.1f%%), fill passes %.2f ms; rl iter %.2f ms" % (d["ms"], 100 * d["frac"], d["fill_passes_ms"], r["roofline"]["ms"]))
PY

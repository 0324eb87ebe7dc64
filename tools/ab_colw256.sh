set -e
mkdir -p gpurun_out/ab
run() {  # name env
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-ops > gpurun_out/ab/$1.json 2> gpurun_out/ab/$1.err || (tail -5 gpurun_out/ab/$1.err; exit 1)
  python - <<PY
import json
r=json.load(open("gpurun_out/ab/$1.json"))
print("$1".ljust(24), "ms/step %.1f"%r["ms_per_step"], "rl iter ms %.2f"%r["roofline"]["ms"])
PY
}
unset BHCORE_LIB; export BH_FC_COLW=4; run default_auto
export BH_FC_COLW=1; run default_all
export BHCORE_LIB=$PWD/biahub_amd/build/variants/libbhcore_colw256.so
export BH_FC_COLW=1; run nt256_all
export BH_FC_COLW=3; run nt256_zonly
export BH_FC_COLW=2; run nt256_yonly

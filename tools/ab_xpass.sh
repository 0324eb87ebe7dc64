set -e
mkdir -p gpurun_out/ab
python -m pytest tests/test_io_cli.py -x -q -m gpu -k "two_ranks or end_to_end or stabiliz" > gpurun_out/ab/cli_tests.log 2>&1 || (tail -40 gpurun_out/ab/cli_tests.log; exit 1)
tail -3 gpurun_out/ab/cli_tests.log
for v in default xnt512 xnt512r16; do
  if [ $v = default ]; then unset BHCORE_LIB; else export BHCORE_LIB=$PWD/biahub_amd/build/variants/libbhcore_$v.so; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-ops > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || (tail -5 gpurun_out/ab/$v.err; exit 1)
  python - <<PY
import json
r=json.load(open("gpurun_out/ab/$v.json"))
print("$v", "ms/step %.1f"%r["ms_per_step"], "rl iter ms %.2f"%r["roofline"]["ms"], "deskew %.2f fill %.2f"%(r["roofline_deskew"]["ms"], r["roofline_deskew"]["fill_passes_ms"]))
PY
done

python -m pytest tests -m gpu -x -q > gpurun_out/r03d_tests.log 2>&1; tail -3 gpurun_out/r03d_tests.log
python bench.py > gpurun_out/r03d_bench_default.json 2> gpurun_out/r03d_bench_default.err; tail -c 600 gpurun_out/r03d_bench_default.json
(python tools/plate_bench.py --order deskew-first; python tools/plate_bench.py; python tools/plate_bench.py --deconv tikhonov) > gpurun_out/r03d_plate.log 2>&1; cut -c1-420 gpurun_out/r03d_plate.log

set -e
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -x -q -k "split_batch or sharding" > gpurun_out/r03/test_split.log 2>&1 || { tail -30 gpurun_out/r03/test_split.log; exit 1; }
python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_like.json 2> gpurun_out/r03/bench_driver_like.err || { tail -30 gpurun_out/r03/bench_driver_like.err; exit 1; }
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err || { tail -30 gpurun_out/r03/bench_default.err; exit 1; }

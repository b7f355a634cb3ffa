set -e
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/gpu_tests.log 2>&1 || { tail -60 gpurun_out/r03/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r03/gpu_tests.log

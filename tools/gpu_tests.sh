# GPU box: run (a subset of) the -m gpu tests into gpurun_out/ROUND/NAME.log.  Usage: bash tools/gpu_tests.sh ROUND NAME [pytest args ...]
set -e
RND=${1:-r04}; NAME=${2:-gpu_tests}; shift 2 || true
mkdir -p gpurun_out/$RND
python -m pytest tests -m gpu -x -q "$@" > gpurun_out/$RND/$NAME.log 2>&1 || { tail -60 gpurun_out/$RND/$NAME.log; exit 1; }
tail -3 gpurun_out/$RND/$NAME.log

# A/B of build variants of the env translation unit (GPU box; the variants are built beforehand, e.g. in the container, as
# pioneer_amd/csrc/libpioneer_amd_NAME.so).  Usage: bash tools/env_ab.sh ROUND TAG NAME [NAME ...]
# Runs bench.py's main leg alternately (default, variants, default, ...) three times; output gpurun_out/ROUND/env_ab_TAG.jsonl
set -e
R=$GRAFT_REPO_ROOT; RND=$1; TAG=$2; shift 2
O=$R/gpurun_out/$RND; mkdir -p $O
MAIN="$R/bench.py --steps 2000 --warmup 100 --fused-leg 0 --large-envs 0 --dynamic-leg 0 --split-leg 0 --ppo-iters 0 --no-cpu-baseline"
for rep in 1 2 3; do
  for lib in default "$@"; do
    if [ "$lib" != default ]; then export PNR_LIB_PATH=$R/pioneer_amd/csrc/libpioneer_amd_$lib.so; else unset PNR_LIB_PATH; fi
    python3 $MAIN 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'lib':'$lib','rep':$rep,'value':d['value'],'avg_launch_ms':d['roofline']['avg_launch_ms'],'frac':d['roofline']['frac']}))" >> $O/env_ab_$TAG.jsonl
  done
done
cat $O/env_ab_$TAG.jsonl

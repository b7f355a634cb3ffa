set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
for rep in 1 2; do
for lib in default "$@"; do
  if [ "$lib" != default ]; then export PNR_LIB_PATH=$R/pioneer_amd/csrc/$lib; else unset PNR_LIB_PATH; fi
  for n in 65536 8192; do
    python $R/bench.py --envs $n --no-cpu-baseline --ppo-iters 0 --steps 2000 --warmup 200 --dynamic-leg 0 --large-envs 0 --fused-leg 32 --split-leg 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', $n, 'us/step %.3f' % (d['roofline']['avg_launch_ms']*1e3), 'frac %.3f' % d['roofline']['frac'], 'rollout us/step %.3f' % (d['fused_rollout']['avg_launch_ms']*1e3/32))"
  done
done
done

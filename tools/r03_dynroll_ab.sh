set -e
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in default "$@"; do
  if [ "$lib" != default ]; then export PNR_LIB_PATH=$R/pioneer_amd/csrc/$lib; else unset PNR_LIB_PATH; fi
  python $R/bench.py --mode dynamic --randomize --gravity 9.81 --fused 32 --envs 65536 --no-cpu-baseline --ppo-iters 0 --steps 640 --warmup 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib rollout us/step %.3f' % (d['roofline']['avg_launch_ms']*1e3/32), 'value %.3e' % d['value'])"
done
done

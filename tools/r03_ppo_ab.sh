# usage: bash tools/r03_ppo_ab.sh [N]  -- the bench's ppo_loop leg alone, N times (default 3)
set -e
A="--steps 200 --fused-leg 0 --large-envs 0 --dynamic-leg 0 --split-leg 0 --no-cpu-baseline --ppo-iters 20"
for i in $(seq ${1:-3}); do python bench.py $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); p=d['ppo_loop']; print(round(p['value']/1e6,2), 'M env-steps/s; learn', round(p['learn_time_s'],4), 's, sample', round(p['sample_time_s'],4), 's over', p['iters'], 'iterations')"; done

"""python tools/eval_policy.py CHECKPOINT [--episodes N] [--gif out.gif] [--mode kinematic|dynamic]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pioneer_amd.evaluate import evaluate

ap = argparse.ArgumentParser()
ap.add_argument("checkpoint")
ap.add_argument("--episodes", type=int, default=3)
ap.add_argument("--max-steps", type=int, default=500)
ap.add_argument("--gif", default=None)
ap.add_argument("--mode", default="kinematic", choices=["kinematic", "dynamic"])
ap.add_argument("--stochastic", action="store_true")
ap.add_argument("--frame-stride", type=int, default=2)
a = ap.parse_args()
res = evaluate(a.checkpoint, a.episodes, a.max_steps, a.gif, mode=a.mode, frame_stride=a.frame_stride,
               deterministic=not a.stochastic)
print(json.dumps(res))

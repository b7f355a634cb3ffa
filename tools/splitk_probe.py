import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pioneer_amd.ppo import ActorCritic, PPOConfig
m = ActorCritic(PPOConfig()).cuda()
x = torch.randn(131072, 137, device="cuda")
def step(amp):
    mean, ls, v = m(x, amp)
    (mean.sum() + ls.sum() + v.sum()).backward()
for amp in (False, True, False, True):
    for _ in range(3): step(amp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(amp)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"amp={amp}: host-issue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")

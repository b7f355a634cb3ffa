#!/usr/bin/env python3
"""Measured accuracy of the MLP kernels per operand precision (planes 1 / 2 / 3) on one 32 768-sample minibatch: heads against a
float64 evaluation of the same float32 weights, gradients (pnr_mlp_train_step's flat bucket) against float32 torch autograd of the
PPO loss, and float32 torch itself against float64 for scale.  Prints one JSON line per precision."""
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from pioneer_amd.mlp import HipMLP  # noqa: E402
from pioneer_amd.ppo import PPOConfig, PPOLearner  # noqa: E402
import test_gpu_mlp as T  # noqa: E402

B, R = 32768, 40000
model, _, obs, _, filt = T.make(B, seed=23, rows=R, with_filter=True)
dev = obs.device
rec = T._record(R, dev)
perm = torch.randperm(R, device=dev).contiguous()
rows = perm[:B]
x = T.net_input(obs, rows, filt)
m64 = copy.deepcopy(model).double()
ref_p, ref_v = m64.policy(x.double()), m64.value(x.double())
with torch.no_grad():
    t32 = T.rel(model.policy(x), ref_p)
cfg = PPOConfig(hip_kernels=False, kl_coeff=0.2)
L = PPOLearner(cfg, dev)
L.model.load_state_dict(model.state_dict())
L._ent_c.fill_(0.01); L._kl_c.fill_(0.2)
batch = {"obs": x, **{k: rec[k][rows] for k in ("actions", "mean", "log_std", "logp", "adv", "vtarg", "values")}}
loss, _ = L.loss(batch)
loss.backward()
params_l = [p for net in (L.model.policy, L.model.value) for l in net if isinstance(l, torch.nn.Linear) for p in (l.weight, l.bias)]
# .. and the same loss differentiated in float64: what both float32 autograd and the kernels approximate
L64 = PPOLearner(cfg, dev)
L64.model.load_state_dict(model.state_dict())
L64.model.double()
L64._ent_c.fill_(0.01); L64._kl_c.fill_(0.2)
try:
    loss64, _ = L64.loss({k: v.double() for k, v in batch.items()})
    loss64.backward()
    params_64 = [p.grad for net in (L64.model.policy, L64.model.value) for l in net if isinstance(l, torch.nn.Linear) for p in (l.weight, l.bias)]
    t32_grad = max(T.rel(a.grad.double(), b) for a, b in zip(params_l, params_64))
except Exception as e:                                            # (the loss helper is float32-only somewhere: report and go on)
    params_64, t32_grad = None, repr(e)
klc, entc = torch.tensor(0.2, device=dev), torch.tensor(0.01, device=dev)
for planes in (1, 2, 3):
    mlp = HipMLP(copy.deepcopy(model), B, dev, planes=planes)
    mlp.pack()
    head = mlp.forward_nograd(obs, rows, filt)
    g = mlp.gather_epoch(obs, perm, filt, rec)
    flat = torch.zeros(int(mlp.lib.pnr_mlp_grad_floats()), device=dev)
    means = torch.zeros(8, device=dev)
    xs = g["xs"][:B] if planes == 1 else g["xs"][:, :B]
    mlp.train_step(None, None, None, {k: g[k][:B] for k in mlp.REC_KEYS}, klc, entc, 0.3, 10.0, 1.0, means, 1e-3, flat_grad=flat, xs_in=xs)
    gerr = [T.rel(a, b.grad) for a, b in zip(T._unpack_flat(flat), params_l)]
    gerr64 = max(T.rel(a.double(), b) for a, b in zip(T._unpack_flat(flat), params_64)) if params_64 else None
    print(json.dumps({"planes": planes, "heads_policy_rel_l2_vs_float64": T.rel(head[0, :, :12], ref_p), "heads_value_rel_l2_vs_float64": T.rel(head[1, :, :1], ref_v),
                      "torch_float32_heads_rel_l2_vs_float64": t32, "gradient_rel_l2_vs_float32_autograd_max": max(gerr),
                      "gradient_rel_l2_vs_float64_autograd_max": gerr64, "torch_float32_gradient_rel_l2_vs_float64_autograd_max": t32_grad,
                      "gradient_rel_l2_by_parameter": dict(zip(T.NAMES[:6], [round(e, 9) for e in gerr[:6]])),
                      "loss_abs_diff": abs(float(means[4]) - float(loss))}))

#!/usr/bin/env python3
"""Where a chunk of the glds ring of mlp_wgrad_kernel goes (diagnostic -DPNR_MLP_STAMPS=1 build; stamps parked in LDS): for chunks 4..7
of every workgroup the cycles from the top of the chunk to [its pieces landed | barrier passed | chunk c + 2 requested | next top =
multiplied], per role; plus workgroup lifetimes.  PNR_LIB_PATH=.../libpioneer_amd_stamps.so python tools/wgrad_ring_stamps.py [OUT.json]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pioneer_amd import _lib  # noqa: E402
from pioneer_amd.mlp import HipMLP  # noqa: E402
from pioneer_amd.ppo import ActorCritic, PPOConfig, gaussian_logp  # noqa: E402

B = 32768
dev = torch.device("cuda", 0)
raw = C.CDLL(_lib.LIB_PATH)
if not hasattr(raw, "pnr_mlp_set_wgrad_stamp_buffer"):
    sys.exit("this library was not built with -DPNR_MLP_STAMPS=1")
torch.manual_seed(0)
PLANES = int(os.environ.get("PLANES", "1"))      # 1: bf16 operands; 2, 3: split float32 operands
mlp = HipMLP(ActorCritic(PPOConfig()).to(dev), B, dev, planes=PLANES)
mlp.pack()
R = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
act, mean, ls = R(B, 6), 0.1 * R(B, 6), 0.1 * R(B, 6)
rec = {"actions": act, "mean": mean, "log_std": ls, "logp": gaussian_logp(act, mean, ls), "values": R(B), "adv": R(B), "vtarg": R(B)}
x32 = R(B, 144)
x32[:, 137:] = 0
planes_of_x = []
for _ in range(PLANES):
    planes_of_x.append(x32.bfloat16())
    x32 = x32 - planes_of_x[-1].float()
xs = planes_of_x[0].contiguous() if PLANES == 1 else torch.stack(planes_of_x).contiguous()
klc, entc, means = torch.tensor(0.2, device=dev), torch.tensor(0.01, device=dev), torch.zeros(8, device=dev)
st = torch.zeros((2, 4, 32, 8, 26), dtype=torch.int64, device=dev)
raw.pnr_mlp_set_wgrad_stamp_buffer(C.c_void_p(st.data_ptr()))
for _ in range(10):
    mlp.train_step(None, None, None, rec, klc, entc, 0.3, 10.0, 1.0, means, 2e-5, xs_in=xs)
torch.cuda.synchronize()
full = st.cpu().numpy().astype(np.int64)
out = {"roles": {}}
t_first = full[..., 0, 24][full[..., 0, 24] > 0].min()
for role, nm in enumerate(("dW2 half 0", "dW2 half 1", "dW1 half 0 + partial sums", "dW1 half 1 + partial sums")):
    r = full[:, role]                       # [nets, slices, waves, 26]
    life = (r[..., 0, 25] - r[..., 0, 24]) / 100.0
    ph = {"lifetime_us_median": float(np.median(life)), "ends_us_after_launch_start": float((r[..., 0, 25].max() - t_first) / 100.0),
          "prologue_to_loop_cycles": int(np.median(r[..., 1] - r[..., 0])), "loop_cycles": int(np.median(r[..., 20] - r[..., 1])),
          "after_loop_cycles": int(np.median(r[..., 22] - r[..., 20]))}
    seg = {"wait_for_pieces": [], "barrier": [], "request_next": [], "multiply": []}
    for k in range(4):
        b = 2 + 4 * k
        seg["wait_for_pieces"].append(r[..., b + 1] - r[..., b])
        seg["barrier"].append(r[..., b + 2] - r[..., b + 1])
        seg["request_next"].append(r[..., b + 3] - r[..., b + 2])
        nxt = r[..., b + 4] if k < 3 else None
        if nxt is not None:
            seg["multiply"].append(nxt - r[..., b + 3])
    for k, v in seg.items():
        a = np.concatenate([x.reshape(-1) for x in v])
        ph[k + "_cycles_median_p10_p90"] = [int(np.median(a)), int(np.percentile(a, 10)), int(np.percentile(a, 90))]
    out["roles"][nm] = ph
    print(nm, json.dumps(ph))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

#!/usr/bin/env python3
"""Where the dW2 roles of mlp_wgrad_kernel spend their cycles (diagnostic -DPNR_MLP_STAMPS=1 build): s_memtime at the start, at the top of
every chunk (behind its first barrier), after the loop and after the slab stores.  Caveat: a stamp is a global store, and the
staging code's s_waitcnt vmcnt(0) waits for it like for any other vector-memory operation — the per-chunk times are upper bounds
(the unstamped kernel's workgroups live shorter); finer stamps inside a chunk measured mostly their own acknowledgements and were removed.
  PNR_LIB_PATH=.../libpioneer_amd_stamps.so python tools/wgrad_stamps.py [OUT.json]"""
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
mlp = HipMLP(ActorCritic(PPOConfig()).to(dev), B, dev)
mlp.pack()
R = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
act, mean, ls = R(B, 6), 0.1 * R(B, 6), 0.1 * R(B, 6)
rec = {"actions": act, "mean": mean, "log_std": ls, "logp": gaussian_logp(act, mean, ls), "values": R(B), "adv": R(B), "vtarg": R(B)}
xs = R(B, 144).bfloat16().contiguous()
klc, entc, means = torch.tensor(0.2, device=dev), torch.tensor(0.01, device=dev), torch.zeros(8, device=dev)
st = torch.zeros((2, 4, 32, 8, 26), dtype=torch.int64, device=dev)
raw.pnr_mlp_set_wgrad_stamp_buffer(C.c_void_p(st.data_ptr()))
for _ in range(10):
    mlp.train_step(None, None, None, rec, klc, entc, 0.3, 10.0, 1.0, means, 2e-5, xs_in=xs)
torch.cuda.synchronize()
full = st.cpu().numpy().astype(np.int64)
t_first = full[..., 0, 24][full[..., 0, 24] > 0].min()
for role, nm in enumerate(("dW2, input columns 0..127", "dW2, input columns 128..255", "dW1 + db1 rows 0..127, half of the layer-3 partial sums", "dW1 + db1 rows 128..255, the other half")):
    r = full[:, role]
    life = (r[..., 0, 25] - r[..., 0, 24]) / 100.0
    print(f"role {role} ({nm}): a workgroup lives {np.median(life):.1f} us (p90 {np.percentile(life, 90):.1f}); last one ends {(r[..., 0, 25].max() - t_first) / 100.0:.1f} us after the first workgroup of the launch started")
s = full[:, :2]          # the two dW2 roles
rt = (s[..., 25] - s[..., 24]) / 100.0
print(f"a dW2 workgroup lives {np.median(rt):.1f} us (median; s_memrealtime), the launch spans {(s[..., 25].max() - s[..., 24].min()) / 100.0:.1f} us")
names = ["first chunk: request, wait for it"] + [f"chunk {k}: stage, barrier, request the next, multiply, barrier" for k in range(15)] + ["chunk 15 (nothing to request)", "slab stores"]
idx = list(range(0, 17)) + [20, 22]
out = {"phases": []}
for k in range(len(idx) - 1):
    d = s[..., idx[k + 1]] - s[..., idx[k]]
    row = {"phase": names[k], "cycles_median_p10_p90": [int(np.median(d)), int(np.percentile(d, 10)), int(np.percentile(d, 90))]}
    out["phases"].append(row)
    print(f"{names[k]:44s} {row['cycles_median_p10_p90'][0]:8d} {row['cycles_median_p10_p90'][1]:7d} {row['cycles_median_p10_p90'][2]:7d}")
print("whole:", int(np.median(s[..., 22] - s[..., 0])), "cycles")
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

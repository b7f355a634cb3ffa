#!/usr/bin/env python3
"""Where mlp_train_kernel (the weight-stationary fused learner kernel) spends its cycles: pnr_mlp_train_step at 32 768 samples
on the diagnostic variant built with -DPNR_MLP_STAMPS=1.  Every wave stamps s_memtime at the phase boundaries of its workgroup's
SECOND tile, after each slot's barrier (plus kernel start, end of the weight preload, kernel end); printed per phase: median / p10 / p90 over the waves.
  PNR_LIB_PATH=.../libpioneer_amd_stamps.so python tools/mlp_train_stamps.py OUT.json [B]"""
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

NAMES = {(0, 1): "preload: weights, biases, first input", (2, 3): "s1  P1 (both halves) | dZ1 epilogue of the tile before, half 1",
         (3, 4): "s2  tanh1 half 0, dZ1 store of the tile before", (4, 5): "s3  P2 half 0 | tanh1 half 1, record request",
         (5, 6): "s4  P2 half 1 | tanh2 half 0, H1 store", (6, 7): "s5  head half 0 (waves 0, 1) | tanh2 half 1, record park",
         (7, 8): "s6  head half 1 (waves 4, 5) | loss half 0 (waves 0-3), H2 store", (8, 9): "s7  dH2 + dZ2 epilogue half 0 | loss half 1 (waves 4-7)",
         (9, 10): "s8  P5 half 0 | dH2 + dZ2 epilogue half 1, requests", (10, 11): "s9  P5 half 1 | dZ1 epilogue half 0, dZ2 store, input park",
         (2, 11): "the whole tile", (0, 22): "the whole kernel"}

B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dev = torch.device("cuda", 0)
lib = _lib.load_library()
raw = C.CDLL(_lib.LIB_PATH)
if not hasattr(raw, "pnr_mlp_set_stamp_buffer"):
    sys.exit("this library was not built with -DPNR_MLP_STAMPS=1")
torch.manual_seed(0)
model = ActorCritic(PPOConfig()).to(dev)
mlp = HipMLP(model, B, dev)
mlp.pack()
R = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
act, mean, ls = R(B, 6), 0.1 * R(B, 6), 0.1 * R(B, 6)
rec = {"actions": act, "mean": mean, "log_std": ls, "logp": gaussian_logp(act, mean, ls), "values": R(B), "adv": R(B), "vtarg": R(B)}
xs = R(B, 144).bfloat16().contiguous()
klc, entc, means = torch.tensor(0.2, device=dev), torch.tensor(0.01, device=dev), torch.zeros(8, device=dev)
cus = torch.cuda.get_device_properties(dev).multi_processor_count
per_net = min((B + 63) // 64, cus // 2)
wgs = per_net * 2
stamps = torch.zeros((wgs, 8, 26), dtype=torch.int64, device=dev)
raw.pnr_mlp_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
for _ in range(10):
    mlp.train_step(None, None, None, rec, klc, entc, 0.3, 10.0, 1.0, means, 2e-5, xs_in=xs)
torch.cuda.synchronize()
full = stamps.cpu().numpy().astype(np.int64)
rt0, rt1 = full[:, 0, 24], full[:, 0, 25]
span_us = (rt1.max() - rt0.min()) / 100.0
wg_us = (rt1 - rt0) / 100.0
clk = (full[:, 0, 22] - full[:, 0, 0]) / np.maximum(wg_us, 1e-9) / 1e3
out = {"workgroups": wgs, "tiles_per_workgroup": (B + 63) // 64 / per_net, "launch_span_us": float(span_us), "workgroup_us_median": float(np.median(wg_us)),
       "in_kernel_clock_ghz_median": float(np.median(clk)), "phases": []}
print(f"launch span {span_us:.1f} us, a workgroup lives {out['workgroup_us_median']:.1f} us (median) at {out['in_kernel_clock_ghz_median']:.2f} GHz")
print(f"{'phase (second tile of each workgroup)':52s} {'median':>8s} {'p10':>7s} {'p90':>7s}   waves 0-3 / 4-7 medians")
for (i, j), name in NAMES.items():
    d = (full[:, :, j] - full[:, :, i])
    lo, hi = d[:, :4].reshape(-1), d[:, 4:].reshape(-1)
    row = {"phase": name, "cycles_median_p10_p90": [int(np.median(d)), int(np.percentile(d, 10)), int(np.percentile(d, 90))],
           "waves0_3_median": int(np.median(lo)), "waves4_7_median": int(np.median(hi))}
    out["phases"].append(row)
    print(f"{name:52s} {row['cycles_median_p10_p90'][0]:8d} {row['cycles_median_p10_p90'][1]:7d} {row['cycles_median_p10_p90'][2]:7d}   {row['waves0_3_median']} / {row['waves4_7_median']}")
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

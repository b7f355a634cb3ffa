#!/usr/bin/env python3
"""Where a tile of the fused learner kernel (mlp_forward_kernel<true>) spends its cycles: run pnr_mlp_train_step at 32 768
samples on the diagnostic variant built with -DPNR_MLP_STAMPS=1 (every wave stamps s_memtime at its phase boundaries into a
buffer of its own) and print, per phase, the median / p10 / p90 over all workgroups of the LAST launch, per wave 0 and waves
1-3.  Build + run (GPU box):
  python -c "from pioneer_amd import _lib, os; ..."  (see tools/r03_mlp_ab.sh) ; PNR_LIB_PATH=.../libpioneer_amd_stamps.so python tools/mlp_stamps.py OUT.json"""
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

NAMES = ["prefetch W1", "stage 0 (tile copy, barriers)", "layer-1 product", "layer-1 epilogue (tanh)", "barrier", "(unused slot)",
         "layer-2 product", "barrier", "layer-2 epilogue (tanh)", "barrier", "(unused slot)", "head product", "W3T fetch + barrier",
         "loss (all waves)", "barrier after loss", "layer-3 partial products + dH2 product + epilogue", "W2T prefetch + barrier", "dZ2 store + W2T product",
         "(unused slot)", "barrier", "(unused slot)", "dZ1 epilogue", "barrier + dZ1 store issue"]

B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dev = torch.device("cuda", 0)
lib = _lib.load_library()
raw = C.CDLL(_lib.LIB_PATH)
if not hasattr(raw, "pnr_mlp_set_stamp_buffer"):
    sys.exit("this library was not built with -DPNR_MLP_STAMPS=1")
torch.manual_seed(0)
model = ActorCritic(PPOConfig()).to(dev)
PLANES = int(os.environ.get("PLANES", "1"))      # 1: bf16 operands; 2, 3: split float32 operands
mlp = HipMLP(model, B, dev, planes=PLANES)
mlp.pack()
R = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
act, mean, ls = R(B, 6), 0.1 * R(B, 6), 0.1 * R(B, 6)
rec = {"actions": act, "mean": mean, "log_std": ls, "logp": gaussian_logp(act, mean, ls), "values": R(B), "adv": R(B), "vtarg": R(B)}
x32 = R(B, 144)
x32[:, 137:] = 0
planes_of_x = []
for _ in range(PLANES):                       # the split planes of the float32 input: p0 = bf16(x), p1 = bf16(x - p0), ...
    planes_of_x.append(x32.bfloat16())
    x32 = x32 - planes_of_x[-1].float()
xs = planes_of_x[0].contiguous() if PLANES == 1 else torch.stack(planes_of_x).contiguous()
klc, entc, means = torch.tensor(0.2, device=dev), torch.tensor(0.01, device=dev), torch.zeros(8, device=dev)
wgs = (B // 64) * 2
stamps = torch.zeros((wgs, 8, 26), dtype=torch.int64, device=dev)
raw.pnr_mlp_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
for _ in range(10):
    mlp.train_step(None, None, None, rec, klc, entc, 0.3, 10.0, 1.0, means, 2e-5, xs_in=xs)
torch.cuda.synchronize()
full = stamps.cpu().numpy().astype(np.int64)
st = full[:, :, :23]
rt0, rt1 = full[:, 0, 24], full[:, 0, 25]                     # s_memrealtime (100 MHz) at start / end of wave 0
d = np.diff(st, axis=2)                       # [wgs, 4, 22] cycles per phase
span_us = (rt1.max() - rt0.min()) / 100.0
tile_us = (rt1 - rt0) / 100.0
clk = (st[:, 0, 22] - st[:, 0, 0]) / np.maximum(tile_us, 1e-9) / 1e3       # GHz: shader cycles per microsecond of real time
start_us = (rt0 - rt0.min()) / 100.0
out = {"workgroups": wgs, "launch_span_us": float(span_us), "tile_cycles_median": int(np.median(st[:, :, 22] - st[:, :, 0])),
       "tile_us_median": float(np.median(tile_us)), "in_kernel_clock_ghz_median": float(np.median(clk)),
       "tile_start_us_percentiles_10_50_75_90_100": [float(np.percentile(start_us, q)) for q in (10, 50, 75, 90, 100)], "phases": []}
print(f"launch span {span_us:.1f} us; a tile takes {out['tile_cycles_median']} cycles = {out['tile_us_median']:.1f} us (median) at "
      f"{out['in_kernel_clock_ghz_median']:.2f} GHz; tile start times (us) p10/50/75/90/100: {out['tile_start_us_percentiles_10_50_75_90_100']}")
print(f"{'phase':44s} {'wave0 med':>9s} {'p10':>7s} {'p90':>7s} | {'others med':>9s} {'p10':>7s} {'p90':>7s}")
for i in range(22):
    a, b = d[:, 0, i], d[:, 1:, i].reshape(-1)
    row = {"phase": NAMES[i + 1], "wave0": [int(np.median(a)), int(np.percentile(a, 10)), int(np.percentile(a, 90))],
           "waves1_3": [int(np.median(b)), int(np.percentile(b, 10)), int(np.percentile(b, 90))]}
    out["phases"].append(row)
    print(f"{NAMES[i + 1]:44s} {row['wave0'][0]:9d} {row['wave0'][1]:7d} {row['wave0'][2]:7d} | {row['waves1_3'][0]:9d} {row['waves1_3'][1]:7d} {row['waves1_3'][2]:7d}")
# when do workgroups start?  (dispatch order = linear block id: x fastest, then net)
order = np.arange(wgs)
out["tile_start_us_every_5_percent"] = [round(float(np.percentile(start_us, q)), 2) for q in range(0, 101, 5)]
out["tile_start_us_by_block_id_blocks_of_64"] = [round(float(np.median(start_us[i:i + 64])), 2) for i in range(0, wgs, 64)]
out["tile_end_us_every_5_percent"] = [round(float(np.percentile((rt1 - rt0.min()) / 100.0, q)), 2) for q in range(0, 101, 5)]
print("start times (us), every 5 %:", out["tile_start_us_every_5_percent"])
print("median start time (us) of blocks 0-63, 64-127, ...:", out["tile_start_us_by_block_id_blocks_of_64"])
print("end times (us), every 5 %:", out["tile_end_us_every_5_percent"])
# where do workgroups run?  slot 23 = XCC_ID << 32 | HW_ID of wave 0 (gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13)
hw = full[:, 0, 23]
xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
cu_key = (xcc << 16) | (hwid & 0xFF00)                 # XCC, SE, SH, CU
end_us = (rt1 - rt0.min()) / 100.0
by_cu = {}
for i in range(wgs):
    by_cu.setdefault(int(cu_key[i]), []).append((float(start_us[i]), float(end_us[i]), i))
per_cu = sorted(len(v) for v in by_cu.values())
out["compute_units_seen"] = len(by_cu)
out["workgroups_per_cu_min_median_max"] = [per_cu[0], per_cu[len(per_cu) // 2], per_cu[-1]]
# per CU: its workgroups in start order; first-round pair = the two that start first
first_fast, first_slow, last_end = [], [], []
for v in by_cu.values():
    v.sort()
    if len(v) >= 2:
        a, b = v[0], v[1]
        d0, d1 = a[1] - a[0], b[1] - b[0]
        first_fast.append(min(d0, d1)); first_slow.append(max(d0, d1))
    last_end.append(max(x[1] for x in v))
out["first_round_pair_on_a_cu_us_fast_slow_median"] = [float(np.median(first_fast)), float(np.median(first_slow))]
out["last_end_per_cu_us_percentiles_0_10_50_90_100"] = [float(np.percentile(last_end, q)) for q in (0, 10, 50, 90, 100)]
per_xcc = {}
for i in range(wgs):
    per_xcc.setdefault(int(xcc[i]), []).append(float(tile_us[i]))
out["tile_us_median_by_xcc"] = {str(k): round(float(np.median(v)), 2) for k, v in sorted(per_xcc.items())}
out["workgroups_by_xcc"] = {str(k): len(v) for k, v in sorted(per_xcc.items())}
print("CUs seen:", out["compute_units_seen"], "workgroups per CU (min / median / max):", out["workgroups_per_cu_min_median_max"])
print("first-round pair of a CU, tile time (us) of the faster / slower one (medians):", out["first_round_pair_on_a_cu_us_fast_slow_median"])
print("end of the last tile per CU (us), p0/10/50/90/100:", out["last_end_per_cu_us_percentiles_0_10_50_90_100"])
print("tile time by XCC:", out["tile_us_median_by_xcc"], "workgroups by XCC:", out["workgroups_by_xcc"])
early = start_us < 0.25 * span_us
out["tile_us_first_round"] = float(np.median(tile_us[early]))
out["tile_us_later"] = float(np.median(tile_us[~early])) if (~early).any() else None
out["workgroups_first_round"] = int(early.sum())
print(f"tiles that start in the first quarter of the launch ({out['workgroups_first_round']} workgroups): {out['tile_us_first_round']:.1f} us each; later ones: {out['tile_us_later']}")
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

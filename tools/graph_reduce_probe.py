#!/usr/bin/env python3
"""Minimal stand-alone form of the r01/r02 "garbage after the second hipGraph replay" finding: a torch reduction over
the MIDDLE axis of a large [31, 16384, 137] float32 tensor (what MeanStdFilter.observe() did inside the captured sampling
loop), captured once and replayed on fresh inputs.  Prints, per replay, the error against eager and against the PREVIOUS
replay's expected result (a stale output would match that).  Usage: python tools/graph_reduce_probe.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
res = []
for shape in ((31, 16384, 137), (31, 16384, 128), (8, 16384, 137), (31, 4096, 137), (31, 16384, 16)):
    for via_temp in (False, True):
        x = torch.randn(*shape, generator=g, device=dev)
        piv = torch.randn(shape[-1], generator=g, device=dev)
        fn = (lambda t: ((t - piv).sum(1), ((t - piv) * (t - piv)).sum(1))) if via_temp else (lambda t: (t.sum(1), (t * t).sum(1)))
        for _ in range(2):
            fn(x)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            y1, y2 = fn(x)
        prev = None
        rows = []
        for it in range(4):
            x.copy_(torch.randn(*shape, generator=g, device=dev) + it)
            gr.replay()
            torch.cuda.synchronize()
            r1, r2 = fn(x)
            e1 = float((y1 - r1).abs().max() / r1.abs().max())
            e2 = float((y2 - r2).abs().max() / r2.abs().max())
            stale = None if prev is None else float((y1 - prev).abs().max() / prev.abs().max())
            rows.append({"replay": it, "rel_err_sum": e1, "rel_err_sumsq": e2, "rel_diff_to_previous_expected": stale,
                         "finite": bool(torch.isfinite(y1).all() and torch.isfinite(y2).all()), "y_absmax": float(y1.abs().max())})
            prev = r1.clone()
        res.append({"shape": shape, "via_temp": via_temp, "replays": rows})
        print(res[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "graph_reduce_probe.json"), "w"), indent=1)

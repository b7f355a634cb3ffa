#!/usr/bin/env python3
"""Root-cause probe for the r01 "NaNs after hipGraph replay with 1- / 12-wide head GEMMs" finding.

Part A isolates the BLAS library: one bf16 / f32 `linear(x[B,256], w[N,256], b[N])` captured in a hipGraph between
two canary tensors allocated inside the same capture (same private pool, adjacent blocks), replayed on fresh inputs;
the output is compared with an fp32 eager reference and the canaries and a guard band behind the output row block
are checked.  N in {1, 12, 16}.

Part B runs the real trainer (sampling graph + learner graphs, RCCL group of one rank with the multi-rank code
paths on, exactly the r01 failing test's set-up) with the head GEMMs at their natural widths (PNR_PPO_HEAD_PAD=1 in
the environment) or padded (default) and reports, per iteration, the first tensor that is not finite.

Writes gpurun_out/nan_repro_<tag>.json.  Usage: python tools/graph_nan_repro.py [A|B] [tag]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def part_a():
    dev = torch.device("cuda", 0)
    res = []
    for dtype in (torch.bfloat16, torch.float32):
        for B in (2048, 16384):
            for N in (1, 12, 16):
                g = torch.Generator(device=dev).manual_seed(B + N)
                x = torch.randn(B, 256, generator=g, device=dev).to(dtype)
                w = (torch.randn(N, 256, generator=g, device=dev) * 0.05).to(dtype)
                b = torch.randn(N, generator=g, device=dev).to(dtype)
                for _ in range(3):
                    F.linear(x, w, b)                       # library warm-up outside the capture
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    c1 = torch.full((1024,), 7.0, device=dev)
                    out = F.linear(x, w, b)
                    c2 = torch.full((1024,), 9.0, device=dev)
                    h = torch.tanh(F.linear(x, w, b))       # a second use: workspace reuse inside one graph
                bad = 0
                worst = 0.0
                for it in range(20):
                    x.copy_(torch.randn(B, 256, generator=g, device=dev).to(dtype))
                    gr.replay()
                    ref = F.linear(x.float(), w.float(), b.float())
                    err = float((out.float() - ref).abs().max())
                    worst = max(worst, err)
                    ok = bool(torch.isfinite(out).all()) and bool((c1 == 7.0).all()) and bool((c2 == 9.0).all()) \
                        and bool(torch.isfinite(h).all())
                    bad += 0 if ok else 1
                tol = 0.06 if dtype == torch.bfloat16 else 1e-3
                res.append({"dtype": str(dtype), "B": B, "N": N, "bad_replays": bad, "max_abs_err": worst,
                            "within_tol": worst <= tol, "out_ptr_mod_256": out.data_ptr() % 256,
                            "out_bytes": out.numel() * out.element_size()})
                print(res[-1], flush=True)
    return res


def finite_report(tr):
    """Name of the first non-finite tensor among everything the loop keeps, or None."""
    groups = {
        "filter": {"n": tr.filter.n, "mean": tr.filter.mean, "m2": tr.filter.m2, "_dsum": tr.filter._dsum,
                   "_dsq": tr.filter._dsq, "_loc": getattr(tr.filter, "_loc", None), "_inv": getattr(tr.filter, "_inv", None)},
        "params": dict(tr.learner.model.named_parameters()),
        "buf": {k: v for k, v in tr.buf.items() if v.dtype.is_floating_point},
        "misc": {"raw_obs": tr.raw_obs, "xin": tr._xin, "stats.ret": tr.stats.ret},
    }
    bad = []
    for gname, d in groups.items():
        for k, v in d.items():
            if v is None:
                continue
            if not bool(torch.isfinite(v.float()).all()):
                bad.append(f"{gname}.{k}")
    return bad


def part_b():
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from pioneer_amd import dist as pdist
    pdist.is_dist = lambda: dist.is_initialized()
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd import ppo
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    out = {"head_pad": ppo._HEAD_PAD, "iters": []}
    env = PioneerVectorEnv(2048, device=dev, seed=3, engine_config=EngineConfig(max_episode_steps=40))
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=16, num_sgd_iter=3, sgd_minibatch_size=8192, lr=3e-4, seed=3,
                                   amp_bf16=True), use_graph=True)
    if ppo._HEAD_PAD < 16:
        tr.learner.fused_loss = False          # the r01 loop at the time of the finding: torch-op loss on [B,12] / [B,1]
    for it in range(6):
        r = tr.train()
        torch.cuda.synchronize()
        bad = finite_report(tr)
        out["iters"].append({"iter": it, "kl": r["kl"], "total_loss": r["total_loss"], "non_finite": bad,
                             "sampling_graph": tr._graph is not None, "learner_graph": tr.learner._graph is not None})
        print(out["iters"][-1], flush=True)
    env.close()
    dist.destroy_process_group()
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "A"
    tag = sys.argv[2] if len(sys.argv) > 2 else which
    res = part_a() if which == "A" else part_b()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"nan_repro_{tag}.json"), "w"), indent=1)

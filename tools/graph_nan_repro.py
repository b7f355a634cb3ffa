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


def part_c():
    """The backward GEMM shapes of the unpadded heads, each eager and under graph replay against an fp32 reference:
    dx = dy[B,n] @ W[n,256] (K = 1 / 12), dW = bmm(dy[S,Bs,n]^T, x[S,Bs,256]) (M = 12, and 8 = the padded value head)."""
    dev = torch.device("cuda", 0)
    res = []
    B, S = 8192, 16
    for n in (1, 8, 12, 16):
        g = torch.Generator(device=dev).manual_seed(n)
        dy = (torch.randn(B, n, generator=g, device=dev) * 1e-3).bfloat16()
        w = (torch.randn(n, 256, generator=g, device=dev) * 0.05).bfloat16()
        x = torch.randn(B, 256, generator=g, device=dev).bfloat16()
        ref_dx = dy.float() @ w.float()
        ref_dw = dy.float().t() @ x.float()

        def ops():
            dx = dy @ w
            dw = torch.bmm(dy.view(S, B // S, n).transpose(1, 2), x.view(S, B // S, 256)).sum(0, dtype=torch.float32)
            return dx, dw
        for _ in range(3):
            ops()
        torch.cuda.synchronize()
        e_dx, e_dw = ops()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            c1 = torch.full((1024,), 7.0, device=dev)
            g_dx, g_dw = ops()
            c2 = torch.full((1024,), 9.0, device=dev)
        worst = {"eager_dx": 0.0, "eager_dw": 0.0, "graph_dx": 0.0, "graph_dw": 0.0}
        finite = True
        for it in range(10):
            gr.replay()
            torch.cuda.synchronize()
            worst["eager_dx"] = max(worst["eager_dx"], float((e_dx.float() - ref_dx).abs().max()))
            worst["eager_dw"] = max(worst["eager_dw"], float((e_dw - ref_dw).abs().max()))
            worst["graph_dx"] = max(worst["graph_dx"], float((g_dx.float() - ref_dx).abs().max()))
            worst["graph_dw"] = max(worst["graph_dw"], float((g_dw - ref_dw).abs().max()))
            finite = finite and bool(torch.isfinite(g_dx).all()) and bool(torch.isfinite(g_dw).all()) \
                and bool((c1 == 7.0).all()) and bool((c2 == 9.0).all())
        res.append({"n": n, "finite_and_canaries": finite, "ref_dx_max": float(ref_dx.abs().max()),
                    "ref_dw_max": float(ref_dw.abs().max()), **worst})
        print(res[-1], flush=True)
    return res


def part_e():
    """Which captured op goes wrong from the SECOND replay on?  Candidates from the sampling graph's filter moments:
    torch reductions that take the multi-block path (per-output semaphores that the launcher zeroes with a
    hipMemsetAsync before every launch — under capture that becomes a memset NODE), and a bare captured
    hipMemsetAsync.  Each case: capture once, replay 5 times on fresh inputs, compare with eager per replay."""
    import ctypes as C
    dev = torch.device("cuda", 0)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    hip.hipMemsetAsync.restype = C.c_int
    g = torch.Generator(device=dev).manual_seed(0)
    cases = {
        "colsum_30720x137": (lambda x: x.sum(0), (30720, 137)),
        "midsum_15x2048x137": (lambda x: x.sum(1), (15, 2048, 137)),
        "midsum_double_sum": (lambda x: x.sum(1).double().sum(0), (15, 2048, 137)),
        "fullsum_16x2048": (lambda x: x.sum(), (16, 2048)),
        "fullsum_524288x137": (lambda x: x.sum(), (524288, 137)),
        "max_16x2048": (lambda x: x.max(), (16, 2048)),
        "cumsum_16x2048": (lambda x: torch.cumsum(x, 0), (16, 2048)),
        "colsum_2048x137": (lambda x: x.sum(0), (2048, 137)),
    }
    res = []
    for name, (fn, shape) in cases.items():
        x = torch.randn(*shape, generator=g, device=dev)
        for _ in range(2):
            fn(x)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            junk = torch.full((4096,), 3.0, device=dev)          # something else in the pool
            y = fn(x)
            junk2 = y.float() * 2 + junk[:1]                     # small allocations after the reduction
        errs = []
        for it in range(5):
            x.copy_(torch.randn(*shape, generator=g, device=dev) + it)
            gr.replay()
            torch.cuda.synchronize()
            ref = fn(x)
            errs.append(float((y.double() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-30)))
        res.append({"case": name, "rel_err_by_replay": errs})
        print(res[-1], flush=True)
    # a bare memset node: buffer := 0xFF.. eagerly, captured memset to 0, then a captured kernel adds 1
    buf = torch.empty(1024, dtype=torch.int32, device=dev)
    out = torch.empty(1024, dtype=torch.int32, device=dev)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(C.c_void_p(buf.data_ptr()), 0, 4096, C.c_void_p(st))
        out.copy_(buf + 1)
    vals = []
    for it in range(4):
        buf.fill_(7 + it)
        gr.replay()
        torch.cuda.synchronize()
        vals.append([int(out.min()), int(out.max())])
    res.append({"case": "bare_hipMemsetAsync_node", "rc": rc, "out_min_max_by_replay (want [1, 1])": vals})
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
    bad, huge = [], {}
    for gname, d in groups.items():
        for k, v in d.items():
            if v is None:
                continue
            vf = v.float()
            if not bool(torch.isfinite(vf).all()):
                bad.append(f"{gname}.{k}")
            else:
                m = float(vf.abs().max())
                if m > 1e4:
                    huge[f"{gname}.{k}"] = m
    return bad, huge


def install_workspace_clearing():
    """PNR_REPRO_CLEAR_WS=1: drop the BLAS workspaces cached per (handle, stream) before and after every capture, as
    torch._inductor.cudagraph_trees.clear_cublas_manager does, so that each graph allocates its own in its own pool."""
    if os.environ.get("PNR_REPRO_CLEAR_WS") != "1":
        return False
    enter, exit_ = torch.cuda.graph.__enter__, torch.cuda.graph.__exit__

    def enter2(self):
        torch._C._cuda_clearCublasWorkspaces()
        return enter(self)

    def exit2(self, *a):
        r = exit_(self, *a)
        torch._C._cuda_clearCublasWorkspaces()
        return r
    torch.cuda.graph.__enter__, torch.cuda.graph.__exit__ = enter2, exit2
    return True


def part_d():
    """bench.py's r02a failure: two trainers one after the other in one process (padded heads, fused loss, no process
    group): 16 384 envs, T = 32, 4 epochs of 32 768- then 131 072-sample minibatches."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    out = {"clear_ws": install_workspace_clearing(), "legs": []}
    order = [int(x) for x in os.environ.get("PNR_REPRO_MBS", "32768,131072").split(",")]
    for mbs in order:
        env = PioneerVectorEnv(16384, device=dev, seed=0, engine_config=EngineConfig(max_episode_steps=500))
        tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=mbs, amp_bf16=True),
                        use_graph=True)
        leg = {"mbs": mbs, "iters": []}
        for it in range(5):
            r = tr.train()
            torch.cuda.synchronize()
            bad, huge = finite_report(tr)
            leg["iters"].append({"iter": it, "kl": r["kl"], "total_loss": r["total_loss"], "non_finite": bad, "huge": huge})
            print(mbs, leg["iters"][-1], flush=True)
        out["legs"].append(leg)
        env.close()
        del tr
        torch.cuda.empty_cache()
    return out


def part_f():
    """Inside the failing configuration of part D (one trainer, padded heads, 32 768-sample minibatches): after every
    collect() compare the filter's pending accumulators, as the sampling graph left them, with an eager recomputation
    from the very buffers the graph read (the previous raw_obs and buf.raw_obs[:T-1])."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    out = {"iters": []}
    env = PioneerVectorEnv(16384, device=dev, seed=0, engine_config=EngineConfig(max_episode_steps=500))
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=32768, amp_bf16=True),
                    use_graph=True)
    T = tr.cfg.rollout_fragment_length
    for it in range(5):
        prev_raw = tr.raw_obs.clone()
        f = tr.filter
        acc0 = (f._dn.clone(), f._dsum.clone(), f._dsq.clone())
        batch = tr.collect()
        torch.cuda.synchronize()
        x = torch.cat([prev_raw, tr.buf["raw_obs"][:T - 1].reshape(-1, 137)]).double()
        piv = f._pivot.double()
        d = x - piv
        e_dsum, e_dsq = d.sum(0), (d * d).sum(0)
        rec = {"iter": it, "graph": tr._graph is not None, "acc_before": [float(acc0[0]), float(acc0[1].abs().max()), float(acc0[2].abs().max())],
               "dn": float(f._dn), "dn_expected": float(x.shape[0]),
               "pivot_is_prev_raw0": bool(torch.equal(f._pivot, prev_raw[0])),
               "dsum_max_abs_err": float((f._dsum - e_dsum).abs().max()), "dsum_ref_max": float(e_dsum.abs().max()),
               "dsq_max_rel_err": float(((f._dsq - e_dsq).abs() / (e_dsq.abs() + 1.0)).max()),
               "bad_cols_dsum": [int(i) for i in torch.nonzero((f._dsum - e_dsum).abs() > 1e-3 * (e_dsum.abs() + 1.0)).flatten()[:20]],
               "raw_obs_finite": bool(torch.isfinite(tr.buf["raw_obs"]).all()), "raw_obs_absmax": float(tr.buf["raw_obs"].abs().max())}
        # the rest of train(), by hand
        tr.filter.sync()
        steps = batch["obs"].shape[0]
        tr.learner.timesteps_total += steps
        info = tr.learner.update(batch, tr.gen)
        torch.cuda.synchronize()
        tr.iteration += 1
        tr.stats.summarize()
        rec.update(kl=info["kl"], mean_absmax=float(f.mean.abs().max()), m2_finite=bool(torch.isfinite(f.m2).all()))
        out["iters"].append(rec)
        print(rec, flush=True)
    env.close()
    return out


def part_g():
    """Variants of part D's failing configuration (PNR_REPRO_VARIANT): which change makes the garbage go away?
      base            nothing changed (control: garbage in the filter moments from the second replay on)
      eager_observe   filter.observe() runs eagerly after the replay instead of inside the sampling graph
      sync_before     torch.cuda.synchronize() right before every collect()
      global_mode     captures use capture_error_mode="global"
      sampler_only    the learner stays eager
      small_observe   observe() in 32 slices of the rollout buffer (no 280 MB temporaries in the graph's pool)
    """
    variant = os.environ.get("PNR_REPRO_VARIANT", "base")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from pioneer_amd import PioneerVectorEnv, EngineConfig
    from pioneer_amd import ppo
    from pioneer_amd.ppo import PPOConfig, PPOTrainer
    if variant == "global_mode":
        ppo._CAPTURE_MODE = "global"
    env = PioneerVectorEnv(16384, device=dev, seed=0, engine_config=EngineConfig(max_episode_steps=500))
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=32768, amp_bf16=True),
                    use_graph=True)
    if variant == "sampler_only":
        tr.learner.use_graph = False
    real_observe = tr.filter.observe
    T = tr.cfg.rollout_fragment_length
    if variant == "eager_observe":
        prev = torch.empty_like(tr.raw_obs)
        tr.filter.observe = lambda x: None
        real_collect = tr.collect

        def collect():
            prev.copy_(tr.raw_obs)
            b = real_collect()
            real_observe(prev)
            real_observe(tr.buf["raw_obs"][:T - 1])
            return b
        tr.collect = collect
    if variant == "small_observe":
        def observe(x):
            if x.dim() == 3:
                for t in range(x.shape[0]):
                    real_observe(x[t])
            else:
                real_observe(x)
        tr.filter.observe = observe
    if variant == "sync_before":
        real_collect2 = tr.collect

        def collect2():
            torch.cuda.synchronize()
            return real_collect2()
        tr.collect = collect2
    out = {"variant": variant, "iters": []}
    for it in range(6):
        r = tr.train()
        torch.cuda.synchronize()
        bad, huge = finite_report(tr)
        out["iters"].append({"iter": it, "kl": r["kl"], "non_finite": bad[:4], "huge": {k: v for k, v in huge.items() if k not in ("filter.n", "filter._inv")}})
        print(variant, out["iters"][-1], flush=True)
    env.close()
    return out


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
    out = {"head_pad": ppo._HEAD_PAD, "iters": [], "clear_ws": install_workspace_clearing()}
    env = PioneerVectorEnv(2048, device=dev, seed=3, engine_config=EngineConfig(max_episode_steps=40))
    graphs = os.environ.get("PNR_REPRO_GRAPHS", "both")           # both | sampler | learner | none
    tr = PPOTrainer(env, PPOConfig(rollout_fragment_length=16, num_sgd_iter=3, sgd_minibatch_size=8192, lr=3e-4, seed=3,
                                   amp_bf16=True), use_graph=True)
    if graphs in ("learner", "none"):
        tr.use_graph = False
    if graphs in ("sampler", "none"):
        tr.learner.use_graph = False
        tr.learner._split = False
    out["graphs"] = graphs
    if os.environ.get("PNR_REPRO_FUSED_LOSS") == "0" or ppo._HEAD_PAD < 16:
        tr.learner.fused_loss = False          # the r01 loop at the time of the finding: torch-op loss on [B,12] / [B,1]
    for it in range(6):
        r = tr.train()
        torch.cuda.synchronize()
        bad, huge = finite_report(tr)
        out["iters"].append({"iter": it, "kl": r["kl"], "total_loss": r["total_loss"], "non_finite": bad, "huge": huge,
                             "sampling_graph": tr._graph is not None, "learner_graph": tr.learner._graph is not None})
        print(out["iters"][-1], flush=True)
    env.close()
    dist.destroy_process_group()
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "A"
    tag = sys.argv[2] if len(sys.argv) > 2 else which
    res = {"A": part_a, "B": part_b, "C": part_c, "D": part_d, "E": part_e, "F": part_f, "G": part_g}[which]()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"nan_repro_{tag}.json"), "w"), indent=1)

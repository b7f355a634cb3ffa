"""One batch of N envs stepped as ONE pnr_step launch per step, against the same N envs as H handles of N/H envs (global env
ids keep the trajectories identical) whose launches go to H streams and overlap: launch t of one part runs under the
dependent-launch boundary of the others.  Same bytes, same outputs.  Prints one JSON line per configuration.
Usage: python tools/two_stream_probe.py [kinematic|dynamic] [N ...]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pioneer_amd import PioneerVectorEnv, EngineConfig, SimulationConfig, _lib

mode = sys.argv[1] if len(sys.argv) > 1 else "kinematic"
sizes = [int(x) for x in sys.argv[2:]] or [65536]
dev = torch.device("cuda:0")
RING = 32


def run(n, parts, graph, K=2000):
    sim = SimulationConfig(gravity=9.81) if mode == "dynamic" else None
    eng = dict(max_episode_steps=500, auto_reset=True, mode=mode, randomize=(mode == "dynamic"))
    per = n // parts
    envs = [PioneerVectorEnv(per, device=dev, seed=0, env_id_offset=i * per, simulation_config=sim, engine_config=EngineConfig(**eng))
            for i in range(parts)]
    for e in envs:
        e.reset()
    g = torch.Generator(device=dev).manual_seed(1234)
    amax = torch.from_numpy(envs[0].a_max).to(dev)
    acts = (torch.rand(16, n, 6, generator=g, device=dev) * 2 - 1) * amax
    obs = torch.empty(RING, n, 137, device=dev)
    rew = torch.empty(RING, n, device=dev)
    done = torch.empty(RING, n, dtype=torch.uint8, device=dev)
    tr = torch.empty(RING, n, dtype=torch.uint8, device=dev)
    lib = envs[0].lib
    main = torch.cuda.current_stream(dev)
    streams = [main] if parts == 1 else [torch.cuda.Stream(dev) for _ in range(parts)]
    V = C.c_void_p

    def calls_for(i):
        o = i * per
        return [(V(acts[t % 16, o:].data_ptr()), V(obs[t % RING, o:].data_ptr()), V(rew[t % RING, o:].data_ptr()),
                 V(done[t % RING, o:].data_ptr()), V(tr[t % RING, o:].data_ptr())) for t in range(16 * RING // 16 * 1)]
    calls = [calls_for(i) for i in range(parts)]
    m = len(calls[0])

    def launch(k, sps):
        for t in range(k):
            for i in range(parts):
                a, o, r, d, x = calls[i][t % m]
                rc = lib.pnr_step(envs[i]._h, a, o, r, d, x, None, sps[i])
                if rc:
                    _lib.check(rc, envs[i]._h)

    def fork():
        if parts > 1:
            ev = torch.cuda.Event()
            ev.record(main)
            for s in streams:
                s.wait_event(ev)

    def join():
        if parts > 1:
            for s in streams:
                ev = torch.cuda.Event()
                ev.record(s)
                main.wait_event(ev)

    sps = [V(s.cuda_stream) for s in streams]
    if graph:
        launch(m, sps)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(dev)
        with torch.cuda.stream(cap):
            gr.capture_begin()
            capm = torch.cuda.current_stream(dev)
            if parts == 1:
                launch(m, [V(capm.cuda_stream)])
            else:
                ev = torch.cuda.Event(); ev.record(capm)
                for s in streams:
                    s.wait_event(ev)
                launch(m, sps)
                for s in streams:
                    e2 = torch.cuda.Event(); e2.record(s); capm.wait_event(e2)
            gr.capture_end()

        def go(k):
            for _ in range(k // m):
                gr.replay()
    else:
        def go(k):
            fork(); launch(k, sps); join()
    K -= K % m
    go(max(m, 200 - 200 % m))
    torch.cuda.synchronize()
    times = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        go(K)
        e1.record(main)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / K * 1e3)
    times.sort()
    for e in envs:
        e.close()
    return times[len(times) // 2]


PARTS = [int(x) for x in os.environ.get("PROBE_PARTS", "1,2,4").split(",")]
GRAPH = [int(x) for x in os.environ.get("PROBE_GRAPH", "0,1").split(",")]
for n in sizes:
    for parts in PARTS:
        for graph in GRAPH:
            us = run(n, parts, graph)
            bytes_ = (750 if mode == "kinematic" else 842) * n
            print(json.dumps({"mode": mode, "envs": n, "parts": parts, "hip_graph": bool(graph), "us_per_step": us,
                              "env_steps_per_s": n / us * 1e6, "hbm_frac": bytes_ / (us * 1e-6) / 8e12}), flush=True)

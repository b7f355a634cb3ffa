#!/usr/bin/env python3
"""bench.py — env-steps/sec of the Pioneer-arm hot path on N MI355X.

One "step" = one pass of the hot path (BulletEnv.step of the reference: integrate + FK + reward +
TimeLimit + auto-reset + 137-float observation) over the env batch, with synthetic actions already
resident in HBM.  The workload is the configuration BASELINE.json's metric is quoted on: 65 536
Pioneer-arm envs IN TOTAL.  At N=1 they all live on one GPU; at N>1 the env axis is sharded in
contiguous blocks over the ranks (65 536 / N per GPU, global env ids, no data-path collective — envs
are independent), so scaling is "strong" and `value` = 65 536 x K / (max over ranks of the wall time).

Launch: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no torchrun environment the
script starts its own N ranks (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py ...` as a child, before anything touches the GPU) and
relays rank 0's line; started under torch.distributed.run it is one of the ranks.  Rank 0 prints ONE
JSON line.

Extra legs in the same line, never part of `value`: "fused_rollout" (pnr_rollout, T steps per launch),
"large_batch" (262 144 envs per launch), "weak_scaling" at N>1 (65 536 envs PER GPU), "dynamics_randomized"
(BASELINE config[4]), "ppo_loop" (config[2] at N=1: 16 384 envs; config[3] at N>1: 65 536 envs in total,
the full rollout+learn loop on SURVEY 8(d)'s contract: T = 32, 4 epochs of 32 768-sample minibatches,
gradients all-reduced over RCCL), "ppo_loop_large_minibatch" (the same loop with 131 072-sample
minibatches, a labelled variant), "cpu_baseline" at N=1.
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ENV_STEP = 750   # 24 action + 92 state in + 80 state out + 548 obs + 6 reward/flags
STEP_IO_BYTES = 24 + 548 + 6   # per env-step regardless of fusion
STATE_BYTES = 92 + 80          # per env per LAUNCH (a fused rollout keeps state in registers)
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MIN_BLOCK_S = 0.05         # a timed K-step block shorter than this is repeated and the median reported
REPEATS = 5


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=65536, help="envs IN TOTAL (sharded over the ranks)")
    ap.add_argument("--obs-layout", default="env_major", choices=["env_major", "feature_major"])
    ap.add_argument("--action-layout", default="env_major", choices=["env_major", "feature_major"])
    ap.add_argument("--fused", type=int, default=1,
                    help="steps per kernel launch (1 = pnr_step per step; T>1 = pnr_rollout of T steps)")
    ap.add_argument("--ring", type=int, default=32, help="obs ring depth (rollout-buffer slices)")
    ap.add_argument("--large-envs", type=int, default=262144,
                    help="also time pnr_step on this many envs per GPU and report it as \"large_batch\" (0 = skip)")
    ap.add_argument("--weak-envs", type=int, default=65536,
                    help="N>1: also time this many envs PER GPU, reported as \"weak_scaling\" (0 = skip)")
    ap.add_argument("--dynamic-leg", type=int, default=1,
                    help="also time dynamics mode with per-env randomisation (BASELINE config[4]) as \"dynamics_randomized\"")
    ap.add_argument("--graph", type=int, default=0, help="1: replay the per-step launches from a hipGraph")
    ap.add_argument("--fused-leg", type=int, default=32,
                    help="also report the fused pnr_rollout rate with this many steps per launch (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--mode", default="kinematic", choices=["kinematic", "dynamic"])
    ap.add_argument("--randomize", action="store_true",
                    help="dynamics mode: per-env link-mass / friction / damping draws at every reset (BASELINE config[4])")
    ap.add_argument("--gravity", type=float, default=0.0, help="dynamics mode: gravity (reference default 0)")
    ap.add_argument("--ppo-iters", type=int, default=-1,
                    help="also time N iterations of the full rollout+learn PPO loop (BASELINE config[2]/[3]) and report it as "
                         "\"ppo_loop\"; default 10 (0 in dynamics mode); at N>1 the gradients are all-reduced over RCCL")
    ap.add_argument("--ppo-envs", type=int, default=0,
                    help="TOTAL envs of the ppo_loop leg (0: 16 384 at N=1 = config[2]; 65 536 sharded over the ranks at N>1 = config[3])")
    ap.add_argument("--ppo-minibatch", type=int, default=32768, help="GLOBAL sgd_minibatch_size of the ppo_loop leg (SURVEY 8(d) config 3)")
    ap.add_argument("--ppo-large-minibatch", type=int, default=131072,
                    help="also report the loop with this global minibatch size as \"ppo_loop_large_minibatch\" (0 = skip)")
    ap.add_argument("--ppo-timeout", type=float, default=300.0)
    return ap.parse_args(argv)


def self_launch(args, argv):
    """--gpus N > 1 without a torchrun environment: start the N ranks as ONE child process tree, before
    this process has imported torch or touched the GPU, relay rank 0's JSON line and return the child's
    exit code.  Nothing is re-exec'ed: this parent only waits."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "2")
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = 0
    for line in proc.stdout:
        s = line.strip()
        if s.startswith("{") and '"metric"' in s:
            print(s, flush=True)
            lines += 1
        elif s:
            print(s, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: the ranks exited 0 but printed {lines} result lines", file=sys.stderr)
        rc = 4
    return rc


def usable_cores():
    """Host cores this process may really use: affinity capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = min(cores, max(1, q // per))
        except Exception:
            pass
    return max(1, cores)


def cpu_baseline(n_envs, seconds):
    """The CPU oracle (a port, float64) timed on this host's cores on a bounded sample."""
    import ctypes as C_
    import numpy as np
    from oracle import COracle
    from oracle.binding import ORC_REF, _ptr
    limit = usable_cores()
    n = min(n_envs, 65536)
    orc = COracle(n, seed=0, precision=ORC_REF, auto_reset=True, nthreads=1)
    orc.reset(want_obs=False)
    rng = np.random.RandomState(1234)
    acts = [(rng.uniform(-1, 1, size=(n, 6)) * orc.a_max).astype(np.float32) for _ in range(4)]
    obs = np.empty((n, 137)); rew = np.empty(n); done = np.empty(n, np.uint8); tr = np.empty(n, np.uint8)

    def run(k, threads):
        t0 = time.perf_counter()
        for i in range(k):
            a = acts[i & 3]
            orc.lib.orc_step_batch(C_.byref(orc.p), orc._sp(), C_.c_int64(n), C_.c_int64(0),
                                   _ptr(a, C_.c_float), _ptr(obs, C_.c_double), _ptr(rew, C_.c_double),
                                   _ptr(done, C_.c_uint8), _ptr(tr, C_.c_uint8), None, C_.c_int(threads))
        return time.perf_counter() - t0

    # a box may advertise more cores than its share: keep the thread count that is actually fastest
    cands = sorted({c for c in (limit, limit // 2, 16, 8) if 1 <= c <= limit})
    best, best_t = 1, None
    for c in cands:
        run(1, c)
        t = run(3, c) / 3
        if best_t is None or t < best_t:
            best, best_t = c, t
    k = max(5, min(20000, int(seconds / max(best_t, 1e-6))))
    dt = run(k, best)
    # SURVEY 8(d) baseline (ii): the same oracle on ONE thread, 4 096 envs, median of five short runs
    single = None
    try:
        n1 = 4096
        o1 = COracle(n1, seed=0, precision=ORC_REF, auto_reset=True, nthreads=1)
        o1.reset(want_obs=False)
        a1 = (rng.uniform(-1, 1, size=(n1, 6)) * o1.a_max).astype(np.float32)
        ob1 = np.empty((n1, 137)); rw1 = np.empty(n1); dn1 = np.empty(n1, np.uint8); tr1 = np.empty(n1, np.uint8)

        def run1(k):
            t0 = time.perf_counter()
            for _ in range(k):
                o1.lib.orc_step_batch(C_.byref(o1.p), o1._sp(), C_.c_int64(n1), C_.c_int64(0), _ptr(a1, C_.c_float),
                                      _ptr(ob1, C_.c_double), _ptr(rw1, C_.c_double), _ptr(dn1, C_.c_uint8), _ptr(tr1, C_.c_uint8),
                                      None, C_.c_int(1))
            return time.perf_counter() - t0
        t1 = run1(2) / 2
        k1 = max(2, int(0.4 / max(t1, 1e-6)))
        rates = sorted(n1 * k1 / run1(k1) for _ in range(5))
        single = {"value": rates[2], "unit": "env-steps/s", "cores": 1, "sample": f"{n1} envs x {k1} steps, median of 5 runs"}
    except Exception as exc:
        single = {"error": f"{type(exc).__name__}: {exc}"}
    # BASELINE.md plan B1: PyBullet single-process replay, only if pybullet exists on this host
    try:
        pyb = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pybullet_replay.py"),
                                         "--steps", "3000"], capture_output=True, text=True,
                                        timeout=120).stdout.strip().splitlines()[-1])
    except Exception as exc:                       # never fabricate a number
        pyb = {"pybullet": f"replay failed: {type(exc).__name__}"}
    return {"pybullet_single_process": pyb, "single_thread": single, "value": n * k / dt, "unit": "env-steps/s", "cores": best, "kind": "port",
            "sample": f"{n} envs x {k} steps, float64 C oracle (oracle/pnr_oracle.c), OpenMP over envs, "
                      f"{best} threads (host advertises {os.cpu_count()}), {dt:.1f} s"}


def shard_range(total, world, rank):
    """pioneer_amd.dist.shard_range, restated so that the launcher-only dry run needs no GPU library."""
    base, rem = divmod(int(total), int(world))
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def dry_run(args, rank, world):
    """PNR_BENCH_DRYRUN=1: the rank plumbing alone (rendezvous, barriers, max over ranks, rank 0's line)
    with a sleep in place of the kernels — what the CPU test of the self-launch path runs.  The line says
    so (`data: dryrun`, value null): it can never be mistaken for a measurement."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    start, cnt = shard_range(args.envs, world, rank)
    if os.environ.get("PNR_BENCH_DRYRUN_FAIL_RANK") == str(rank):     # test hook: a rank that dies
        sys.exit(7)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(1e-5 * args.steps * (1 + rank))
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    cnts = torch.zeros(world, dtype=torch.int64)
    cnts[rank] = cnt
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnts)
    if rank == 0:
        print(json.dumps({"metric": "env-steps/sec", "value": None, "unit": "env-steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(el) / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                          "data": "dryrun",
                          "config": {"workload": "launcher dry run, no kernels", "total_envs": int(cnts.sum()),
                                     "envs_per_gpu": [int(x) for x in cnts], "parallelism": f"env-shard x{world}"}}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    if os.environ.get("PNR_BENCH_DRYRUN") == "1":
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist
    from pioneer_amd import PioneerVectorEnv, EngineConfig, SimulationConfig, _lib
    from pioneer_amd import dist as pdist

    # one process per GPU; PNR_BENCH_SHARE_GPU=1 (rehearsal on a 1-GPU box) maps every rank to the
    # devices that exist and uses gloo, because RCCL refuses two ranks on one device
    share = os.environ.get("PNR_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
    backend = dist.get_backend() if world > 1 else None

    total_envs = args.envs
    env_start, n = shard_range(total_envs, world, rank)      # this rank's block of the global env axis
    T = max(1, args.fused)
    env = PioneerVectorEnv(n, device=dev, seed=0, env_id_offset=env_start,
                           simulation_config=SimulationConfig(gravity=args.gravity),
                           engine_config=EngineConfig(max_episode_steps=500, auto_reset=True,
                                                      obs_layout=args.obs_layout,
                                                      action_layout=args.action_layout, mode=args.mode,
                                                      randomize=args.randomize))
    env.reset()

    # synthetic inputs, resident in HBM before the timed region
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    amax = torch.from_numpy(env.a_max).to(dev)
    n_act = max(16, T, args.fused_leg if (T == 1 and args.mode == "kinematic") else 0)
    if args.action_layout == "env_major":
        acts = (torch.rand(n_act, n, 6, generator=g, device=dev) * 2 - 1) * amax
    else:
        acts = (torch.rand(n_act, 6, n, generator=g, device=dev) * 2 - 1) * amax[:, None]
    ring = max(args.ring, T)
    ring -= ring % T
    if T == 1 and args.fused_leg > 1 and args.mode == "kinematic":
        ring = max(ring, args.fused_leg)
        ring -= ring % args.fused_leg
    obs = torch.empty((ring,) + tuple(env.obs_shape), dtype=torch.float32, device=dev)
    rew = torch.empty((ring, n), dtype=torch.float32, device=dev)
    done = torch.empty((ring, n), dtype=torch.uint8, device=dev)
    trunc = torch.empty((ring, n), dtype=torch.uint8, device=dev)

    lib, h = env.lib, env._h
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    P = lambda t, i: C.c_void_p(t[i].data_ptr())  # noqa: E731

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_blocks(run, K):
        """EXACTLY K steps between barrier + synchronize on both sides, wall time = max over ranks.  A block
        shorter than MIN_BLOCK_S is repeated (REPEATS blocks in all) and the MEDIAN block is reported.
        Returns (median wall s, median HIP-event ms of this rank, repeats, all wall times)."""
        walls, evs = [], []
        reps = 1
        while len(walls) < reps:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            barrier()
            t0 = time.perf_counter()
            ev0.record(stream)       # torch's current stream == the stream pnr_step launches on
            run(K)
            ev1.record(stream)
            barrier()
            walls.append(max_over_ranks(time.perf_counter() - t0))
            evs.append(ev0.elapsed_time(ev1))
            if len(walls) == 1 and walls[0] < MIN_BLOCK_S:     # the same decision on every rank: walls[] is all-reduced
                reps = REPEATS
        return statistics.median(walls), statistics.median(evs), reps, walls

    def timed(T, K, W):
        """K env-steps (after W warm-up steps) with T steps per kernel launch."""
        if T == 1:
            calls = [(P(acts, i % n_act), P(obs, i % ring), P(rew, i % ring), P(done, i % ring), P(trunc, i % ring))
                     for i in range(math.lcm(n_act, ring))]

            def run_eager(k, spx=sp):
                m = len(calls)
                for i in range(k):
                    a, o, r, d, tr = calls[i % m]
                    rc = lib.pnr_step(h, a, o, r, d, tr, None, spx)
                    if rc:
                        _lib.check(rc, h)

            run = run_eager
            if args.graph:
                # the step loop is launch-bound on the host: capture one pass over the obs ring
                # (len(calls) launches of pnr_step) in a hipGraph and replay it
                m = len(calls)
                run_eager(m)
                torch.cuda.synchronize(dev)
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_):
                    run_eager(m, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))

                def run(k):   # noqa: F811
                    assert k % m == 0, f"--steps/--warmup must be multiples of {m} with --graph"
                    for _ in range(k // m):
                        g_.replay()
                K -= K % m
                W = max(m, W - W % m)
        else:
            nslots = ring // T
            # operand shapes must cover what one launch touches: T action slices, T output slices per slot
            assert acts.shape[0] >= T and nslots >= 1 and nslots * T <= obs.shape[0] == rew.shape[0] == done.shape[0] == trunc.shape[0], \
                (acts.shape, obs.shape, T)
            calls = [(P(acts, 0), P(obs, s * T), P(rew, s * T), P(done, s * T), P(trunc, s * T)) for s in range(nslots)]
            K -= K % T
            W -= W % T

            def run(k):
                for i in range(k // T):
                    a, o, r, d, tr = calls[i % nslots]
                    rc = lib.pnr_rollout(h, T, a, o, r, d, tr, sp)
                    if rc:
                        _lib.check(rc, h)
        run(W)
        el, ev_ms, reps, walls = timed_blocks(run, K)
        return el, ev_ms, K, W, reps, walls

    elapsed, ev_ms, K, W, repeats, walls = timed(T, args.steps, args.warmup)
    value = float(total_envs) * K / elapsed

    fused = None
    if T == 1 and args.fused_leg > 1 and args.mode == "kinematic":
        Tf = min(args.fused_leg, ring)
        fel, fev, fK, _, freps, _ = timed(Tf, max(Tf, args.steps), max(Tf, args.warmup))
        fl_ms = fev / (fK // Tf)
        fb = n * (STEP_IO_BYTES * Tf + STATE_BYTES)
        fused = {"value": float(total_envs) * fK / fel, "unit": "env-steps/s", "steps_per_launch": Tf, "steps": fK,
                 "repeats": freps, "ms_per_step": fel / fK * 1e3, "avg_launch_ms": fl_ms,
                 "algorithmic_bytes_per_launch": fb, "achieved_GBps": fb / (fl_ms * 1e-3) / 1e9,
                 "frac": fb / (fl_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                 "note": "pnr_rollout: same kernel, T steps per launch with open-loop actions; state stays in registers"}

    def side_leg(nl, id_off, k, warm, note, bytes_per_env_step=BYTES_PER_ENV_STEP, sim=None, rollout_T=0, **engine_kw):
        """pnr_step on a separate env batch of nl envs per GPU: wall time is the max over ranks between
        barriers, avg_launch_ms is rank 0's HIP-event timing."""
        lenv = PioneerVectorEnv(nl, device=dev, seed=0, env_id_offset=id_off, simulation_config=sim,
                                engine_config=EngineConfig(max_episode_steps=500, auto_reset=True,
                                                           obs_layout=args.obs_layout, action_layout=args.action_layout,
                                                           **engine_kw))
        lenv.reset()
        lring = 8 if nl >= 131072 else 32
        lacts = (torch.rand((4,) + tuple(lenv.action_shape), generator=g, device=dev) * 2 - 1) * \
            (amax if args.action_layout == "env_major" else amax[:, None])
        lobs = torch.empty((lring,) + tuple(lenv.obs_shape), dtype=torch.float32, device=dev)
        lrew = torch.empty((lring, nl), dtype=torch.float32, device=dev)
        ldone = torch.empty((lring, nl), dtype=torch.uint8, device=dev)
        ltr = torch.empty((lring, nl), dtype=torch.uint8, device=dev)
        lh = lenv._h
        lcalls = [(P(lacts, i % 4), P(lobs, i % lring), P(lrew, i % lring), P(ldone, i % lring), P(ltr, i % lring))
                  for i in range(lring)]

        def lrun(kk):
            for i in range(kk):
                a, o, r, d, tr = lcalls[i % lring]
                rc = lib.pnr_step(lh, a, o, r, d, tr, None, sp)
                if rc:
                    _lib.check(rc, lh)
        lrun(warm)
        el, lev, lreps, _ = timed_blocks(lrun, k)
        lms = lev / k
        roll = None
        if rollout_T > 1 and lring >= rollout_T:
            # the same steps as ONE pnr_rollout launch per rollout_T steps (open-loop actions)
            racts = (torch.rand((rollout_T,) + tuple(lenv.action_shape), generator=g, device=dev) * 2 - 1) * \
                (amax if args.action_layout == "env_major" else amax[:, None])
            rcall = lambda: _lib.check(lib.pnr_rollout(lh, rollout_T, P(racts, 0), P(lobs, 0), P(lrew, 0), P(ldone, 0), P(ltr, 0), sp), lh)  # noqa: E731
            for _ in range(3):
                rcall()
            reps = max(4, k // rollout_T)
            rel, _, rreps, _ = timed_blocks(lambda kk: [rcall() for _ in range(kk)], reps)
            roll = {"steps_per_launch": rollout_T, "steps": reps * rollout_T, "repeats": rreps,
                    "ms_per_step": rel / (reps * rollout_T) * 1e3,
                    "value": float(nl) * world * reps * rollout_T / rel, "unit": "env-steps/s"}
        lenv.close()
        if roll:
            note = note + "; `rollout`: pnr_rollout, the steps looped inside one launch"
        return {**({"rollout": roll} if roll else {}), "envs_per_gpu": nl, "steps": k, "repeats": lreps,
                "avg_launch_ms": lms, "ms_per_step": el / k * 1e3,
                "value": float(nl) * world * k / el, "unit": "env-steps/s",
                "env_steps_per_s_per_gpu": nl / (lms * 1e-3),
                "achieved_GBps": bytes_per_env_step * nl / (lms * 1e-3) / 1e9,
                "frac": bytes_per_env_step * nl / (lms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "note": note}

    id_base = total_envs            # the side legs use global env ids beyond the headline batch
    # the same kernel on a larger batch per launch (amortises the ~3.5 us launch + first-load floor)
    large = None
    if T == 1 and args.large_envs > n and args.mode == "kinematic":
        large = side_leg(args.large_envs, id_base + rank * args.large_envs, 256, 32,
                         "pnr_step, one launch per step, same kernel as `value`; avg_launch_ms/frac from rank 0's HIP events")
    id_base += world * args.large_envs

    # BASELINE config[4]: dynamics mode (ABA + PD + limits, 10 sub-steps) with per-env randomised link
    # masses / friction / damping; parity unpinned (the reference never exercises dynamics).  VALU-bound:
    # `frac` is still quoted against HBM with SURVEY 8(d)'s 842 B per env-step.
    dynamic = None
    if T == 1 and args.mode == "kinematic" and args.dynamic_leg:
        dstart, dcnt = shard_range(total_envs, world, rank)
        dynamic = side_leg(dcnt, id_base + dstart, 512, 64,
                           f"{total_envs} envs in total over {world} rank(s); mode=dynamic, randomize=True, gravity 9.81: "
                           "dyn_step_kernel (sub-steps one env per lane with packed "
                           "fp32 ABA, VALU-bound; then lane pairs finish reward/obs); algorithmic bytes 842 B per env-step",
                           bytes_per_env_step=842, sim=SimulationConfig(gravity=9.81), rollout_T=32, mode="dynamic", randomize=True)
        dynamic["total_envs"] = total_envs
        dynamic["value"] = float(total_envs) / (dynamic["ms_per_step"] * 1e-3)
        # the kernel is VALU-issue-bound, not HBM-bound: instructions from the committed SQ-counter pass, time from this run
        try:
            cnt = json.load(open(os.path.join(ROOT, "profiles", "r02_e_dyn_sq_counters.json")))["kernels"]["dyn_step_kernel<1,1,1,0>"]
            instr = cnt["valu_roofline"]["valu_wave_instructions_per_dispatch"] * (dcnt / 65536.0)
            simds, clock = 256 * 4, 2.4e9
            ach = instr / (dynamic["avg_launch_ms"] * 1e-3)
            dynamic["roofline"] = {"bound": "valu", "achieved": ach / 1e9, "peak": simds * clock / 2 / 1e9, "unit": "G wave-instructions/s",
                                   "frac": ach / (simds * clock / 2),
                                   "frac_of_single_wave_issue": cnt["valu_roofline"]["frac_of_single_wave_issue"],
                                   "source": "VALU wave-instructions per 65 536-env launch from profiles/r02_e_dyn_sq_counters.json (rocprofv3 --pmc "
                                             "SQ_INSTS_VALU, not this run), duration = this run's HIP events; peak = one wave64 VALU instruction "
                                             "per 2 cycles per SIMD at 2.4 GHz; one wave per SIMD can issue one per 4"}
        except Exception:
            pass
        if "rollout" in dynamic:
            dynamic["rollout"]["value"] = float(total_envs) / (dynamic["rollout"]["ms_per_step"] * 1e-3)
    id_base += total_envs

    # weak scaling (the r01 headline): every rank steps its own 65 536-env shard
    weak = None
    if T == 1 and world > 1 and args.mode == "kinematic" and args.weak_envs > 0:
        weak = side_leg(args.weak_envs, id_base + rank * args.weak_envs, max(256, args.steps), 64,
                        f"{args.weak_envs} envs PER GPU on {world} ranks (weak scaling; not the metric's configuration)")
        weak["total_envs"] = args.weak_envs * world

    def ppo_leg(iters, global_mbs):
        """BASELINE config[2] at N=1 (16 384 envs, full rollout + learn loop) and config[3] at N>1
        (65 536 envs in total sharded over the ranks, gradients all-reduced over RCCL/xGMI)."""
        from pioneer_amd.ppo import PPOConfig, PPOTrainer
        total = args.ppo_envs if args.ppo_envs > 0 else (16384 if world == 1 else 65536)
        start, cnt = pdist.shard_range(total, world, rank)
        penv = PioneerVectorEnv(cnt, device=dev, seed=0, env_id_offset=start,
                                engine_config=EngineConfig(max_episode_steps=500, auto_reset=True, mode=args.mode))
        mbs = max(1, min(global_mbs // world, 32 * cnt))      # this rank's share of every global minibatch
        pcfg = PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=mbs, amp_bf16=True)
        tr = PPOTrainer(penv, pcfg, use_graph=True)
        tr.train(); tr.train()                       # warm-up: eager iteration, then the graph-captured one
        barrier()
        tp = time.perf_counter()
        rs = [tr.train() for _ in range(iters)]
        barrier()
        tp = max_over_ranks(time.perf_counter() - tp)
        graphed = {"sampling": tr._graph is not None,
                   "learner": ("hip kernels: pnr_mlp_train_step, 3 launches per update, no graph needed" if tr.learner.hip
                               else ("graph" if tr.learner._graph is not None else "eager")),
                   "learner_split_around_allreduce": bool(tr.learner._split) or (tr.learner.hip and world > 1)}
        finite = all(math.isfinite(float(r[k])) for r in rs for k in ("kl", "total_loss"))
        penv.close()
        del tr
        torch.cuda.empty_cache()
        steps = iters * 32 * total
        return {"value": steps / tp, "unit": "env-steps/s", "total_envs": total, "envs_per_gpu": cnt, "rollout_T": 32,
                "num_sgd_iter": 4, "sgd_minibatch_size": mbs * world, "sgd_minibatch_size_per_rank": mbs,
                "sgd_updates_per_iter": 4 * ((32 * cnt) // mbs), "mlp_dtype": "bf16",
                "hip_graph": graphed, "iters": iters, "losses_finite": finite,
                "grad_allreduce": ({"backend": backend, "world_size": world, "bytes": 4 * 205581,
                                    "per": "minibatch, one flat bucket"} if world > 1 else None),
                "sample_time_s": sum(r["sample_time_s"] for r in rs), "learn_time_s": sum(r["learn_time_s"] for r in rs),
                "note": "full loop: policy MLP 137-256-256 fwd per step, GAE, 4 SGD epochs, obs filter, grad all-reduce"}

    def emit(ppo_loop, ppo_large):
        if rank != 0:
            return
        launches = K // T
        launch_ms = ev_ms / launches
        # per launch: T steps of action/obs/reward/flags traffic + ONE state read and write
        algo_bytes = n * (STEP_IO_BYTES * T + STATE_BYTES)
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                rec = json.load(open(prof))
                key = f"{args.mode}:{args.obs_layout}:{n}:{T}"
                if key in rec:
                    traffic = rec[key].get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/pmc_traffic.json (" + str(rec[key].get("profile", "earlier rocprofv3 --pmc passes")) +
                                      "; looked up, NOT measured in this run)")
            except Exception:
                traffic = None
        kname = "pnr::step_kernel" if args.mode == "kinematic" else "pnr::dyn_step_kernel"
        out = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "repeats": repeats, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{total_envs} Pioneer-arm envs in total ({n} per GPU on {world} GPU(s)), physics-only step "
                                   f"(integrate+FK+reward+TimeLimit(500)+auto-reset+obs[137]), random actions U(-a_max,a_max) resident in HBM",
                       "envs_per_gpu": n, "total_envs": total_envs, "mode": args.mode,
                       "randomize": bool(args.randomize), "gravity": args.gravity,
                       "obs_layout": args.obs_layout, "action_layout": args.action_layout,
                       "steps_per_launch": T, "hip_graph": bool(args.graph and T == 1), "obs_ring_slices": ring,
                       "parallelism": f"env-shard x{world}", "block_wall_s": walls},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "avg_launch_ms": launch_ms,
                         "timing": "HIP events on the launch stream around the timed K-step block / launches (rank 0; median block)"},
        }
        if fused:
            out["fused_rollout"] = fused
        if large:
            out["large_batch"] = large
        if dynamic:
            out["dynamics_randomized"] = dynamic
        if weak:
            out["weak_scaling"] = weak
        if ppo_loop:
            out["ppo_loop"] = ppo_loop
        if ppo_large:
            out["ppo_loop_large_minibatch"] = ppo_large
        if not args.no_cpu_baseline and world == 1:      # contract: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    ppo_loop = ppo_large = None
    rc = 0
    if args.ppo_iters < 0:
        args.ppo_iters = 10 if args.mode == "kinematic" else 0      # ~90 ms of timed work per PPO leg
    if args.ppo_iters > 0:
        # the extra leg must never take the main result down: the line is still printed when it fails, but the
        # process then ends NON-ZERO — exceptions are reported in place (rc 5), and a leg that does not come back
        # (e.g. ranks out of step in a collective) is cut off by a watchdog that prints the line without it and
        # ends the process with rc 3
        import threading
        lock, state = threading.Lock(), {"done": False}

        def on_timeout():
            with lock:
                if state["done"]:
                    return
                state["done"] = True
                emit({"error": f"ppo leg did not finish within {args.ppo_timeout:.0f} s"}, None)
                sys.stdout.flush()
                os._exit(3)
        wd = threading.Timer(args.ppo_timeout, on_timeout)
        wd.daemon = True
        wd.start()
        try:
            ppo_loop = ppo_leg(args.ppo_iters, args.ppo_minibatch)
            if args.ppo_large_minibatch > 0 and args.ppo_large_minibatch != args.ppo_minibatch:
                ppo_large = ppo_leg(args.ppo_iters, args.ppo_large_minibatch)
        except Exception as exc:
            err = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            if ppo_loop is None:
                ppo_loop = err
            else:
                ppo_large = err
            rc = 5
        with lock:
            state["done"] = True
        wd.cancel()
        for leg in (ppo_loop, ppo_large):
            if leg and leg.get("losses_finite") is False:
                rc = 5
    emit(ppo_loop, ppo_large)

    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())

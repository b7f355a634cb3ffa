#!/usr/bin/env python3
"""bench.py — env-steps/sec of the Pioneer-arm hot path on N MI355X.

One "step" = one pass of the hot path (BulletEnv.step of the reference: integrate + FK + reward +
TimeLimit + auto-reset + 137-float observation) over the env batch, with synthetic actions already
resident in HBM.  The workload is the configuration BASELINE.json's metric is quoted on: 65 536
Pioneer-arm envs IN TOTAL.  At N=1 they all live on one GPU; at N>1 the env axis is sharded in
contiguous blocks over the ranks (65 536 / N per GPU, global env ids, no data-path collective — envs
are independent), so scaling is "strong" and `value` = 65 536 x K / (max over ranks of the wall time).

Timed region: after W warm-up steps, blocks of EXACTLY K steps are enqueued back to back (`repeats` of them, until the series
lasts >= 60 ms) between barrier + torch.cuda.synchronize() on both sides; `ms_per_step` = max-over-ranks wall time of the series /
(repeats x K), and HIP events recorded on the launch stream between consecutive blocks give the median block (the roofline's
duration) and the spread.  `--split 2 --graph 1` steps the batch as TWO handles of n/2 envs (global env ids: the same
trajectories bit for bit, tests/test_gpu_parity.py) whose launches are independent, each block replayed from a hipGraph; by
default that form is the `split_streams` leg and the headline is one eager pnr_step launch per step.

Launch: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no torchrun environment the
script starts its own N ranks (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py ...` as a child, before anything touches the GPU) and
relays rank 0's line; started under torch.distributed.run it is one of the ranks.  Rank 0 prints ONE
JSON line.

Extra legs in the same line, never part of `value`: "fused_rollout" (pnr_rollout, T steps per launch),
"large_batch" (262 144 envs per launch), "weak_scaling" at N>1 (65 536 envs PER GPU), "dynamics_randomized"
(BASELINE config[4]), "ppo_loop_f32" (config[2] at N=1: 16 384 envs; config[3] at N>1: 65 536 envs in total,
the full rollout+learn loop on SURVEY 8(d)'s contract: T = 32, 4 epochs of 32 768-sample minibatches,
gradients all-reduced over RCCL — with float32-accurate products, the reference learner's arithmetic: the
credited PPO figure), "ppo_loop" (the same loop with bf16 operands: a reduced-precision variant),
"ppo_loop_65536" (N=1: config[3] whole on one GPU, both precisions), "ppo_loop_bf16x3" (N=1),
"ppo_loop_large_minibatch" (bf16, 131 072-sample minibatches, a labelled variant), "cpu_baseline" at N=1,
"build" (the binary's baked-in fingerprints next to the tree's and the counter passes').
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

LEARNER_TRAFFIC = "learner_pmc_traffic.json"      # profiles/: FETCH/WRITE bytes of the learner kernels per update (tools/learner_traffic.py)
BYTES_PER_ENV_STEP = 750   # 24 action + 92 state in + 80 state out + 548 obs + 6 reward/flags
STEP_IO_BYTES = 24 + 548 + 6   # per env-step regardless of fusion
STATE_BYTES = 92 + 80          # per env per LAUNCH (a fused rollout keeps state in registers)
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MIN_SERIES_S = 0.06        # the timed K-step block is repeated back to back until the series is at least this long ...
MIN_REPEATS = 5            # ... and at least this often; `ms_per_step` = series wall time / (repeats x K)
MAX_REPEATS = 20000
DYN_PACKED_FRAC = 361.0 / 977.0                 # v_pk_*_f32 share of the dynamics sub-step loop's VALU instructions (tools/isa_loop_mix.py)
DYN_COUNTERS = "r05_e_dyn_sq_counters.json"      # tools/dyn_counters_summary.py (VALU instructions per launch of the dynamics kernel)


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
    ap.add_argument("--graph", type=int, default=-1,
                    help="1: each timed K-step block is ONE hipGraph replay; 0: eager launches; -1 (default): 0")
    ap.add_argument("--split", type=int, default=-1,
                    help="main leg: step the rank's envs as this many handles (contiguous blocks, global env ids: same trajectories, "
                         "tests/test_gpu_parity.py) whose launches are independent and overlap; -1 (default): 1 (the two-handle form is "
                         "reported as the `split_streams` leg)")
    ap.add_argument("--split-leg", type=int, default=1, help="also report the two-handle / two-stream form of the main batch as \"split_streams\"")
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
    ap.add_argument("--ppo-f32", type=int, default=1,
                    help="N = 1: also report the PPO loop with three bf16 planes per operand as \"ppo_loop_bf16x3\" (\"ppo_loop_f32\", the "
                         "float32-accurate loop, and \"ppo_loop\", its bf16 reduced-precision variant, are always reported)")
    ap.add_argument("--ppo-65536", type=int, default=1,
                    help="N = 1: also report BASELINE config[3] whole on ONE GPU (65 536 envs, T = 32, 32 768-sample minibatches) at both "
                         "precisions as \"ppo_loop_65536\": the first point of SURVEY 8(d)'s 1/2/4/8 table")
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
    ndev = torch.cuda.device_count()                 # counting devices does not initialise the GPU
    if ndev < 1 or (not share and local_rank >= ndev):
        # a user error, not a GPU fault: say so in one line on every rank that cannot be placed, before any GPU call
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible devices (found {ndev}); rank {rank} (local rank {local_rank}) has "
              "no device" + ("" if share or ndev < 1 else "  [PNR_BENCH_SHARE_GPU=1 rehearses N ranks on the devices that exist, over gloo]"),
              file=sys.stderr, flush=True)
        return 2
    dev_index = local_rank % ndev if share else local_rank
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
    stream = torch.cuda.current_stream(dev)
    from pioneer_amd.config import PioneerKinematicConfig, to_c_config
    kc = _lib.PnrConstants()
    _lib.check(_lib.load_library().pnr_get_constants(to_c_config(PioneerKinematicConfig(), SimulationConfig(), EngineConfig()), kc))
    amax = torch.tensor(list(kc.a_max[:]), dtype=torch.float32, device=dev)      # action_space.high, pioneer_knm_env.py:58, :72
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    V = C.c_void_p

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

    def looked_up(path, key_path):
        """A counter value taken in an EARLIER rocprofv3 --pmc pass (profiles/): returned only while the sources of the env
        kernels are the ones that pass ran on (`csrc_sha16`, pioneer_amd._lib.source_fingerprint); else (None, why)."""
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", path)))
        except Exception as exc:
            return None, f"profiles/{path}: {type(exc).__name__}"
        have, want = doc.get("csrc_sha16"), _lib.source_fingerprint()
        if have != want:
            return None, f"stale: kernel changed since the counter pass (profiles/{path} was taken on sources {have}, this build is {want})"
        node = doc
        for k in key_path:
            if not isinstance(node, dict) or k not in node:
                return None, f"profiles/{path} has no entry {'/'.join(key_path)}"
            node = node[k]
        return node, f"profiles/{path} (rocprofv3 --pmc passes on these sources, {want}; looked up, NOT measured in this run)"

    def dyn_valu_roofline(avg_launch_ms, envs):
        cnt, src = looked_up(DYN_COUNTERS, ["kernels", "dyn_step_kernel<1,1,1,0>", "valu_roofline"])
        if cnt is None:
            return {"bound": "valu", "achieved": None, "frac": None, "source": src}
        instr = cnt["valu_wave_instructions_per_dispatch"] * (envs / 65536.0)
        simds, clock = 256 * 4, 2.4e9
        ach = instr / (avg_launch_ms * 1e-3)
        # the same instructions as fp32 LANE-OPERATIONS: 64 per wave-instruction, a packed one (v_pk_*_f32: 361 of the sub-step loop's 977
        # VALU instructions, tools/isa_loop_mix.py) counted twice, against the guide's vector peak of 157.3 TFLOP/s = 78.6 T lane-operations/s
        # (an FMA lane-operation is two flops)
        lane_ops = instr * 64.0 * (1.0 + DYN_PACKED_FRAC)
        lane = {"achieved": lane_ops / (avg_launch_ms * 1e-3) / 1e12, "peak": 157.3 / 2, "unit": "T fp32 lane-operations/s",
                "frac": lane_ops / (avg_launch_ms * 1e-3) / (157.3e12 / 2), "packed_instruction_fraction": DYN_PACKED_FRAC,
                "note": "packed instructions counted twice.  Reconciles the issue model above with the occupancy A/B (DESIGN_HISTORY r03: 1.75 / 1.73 / "
                        "1.60 us per 65 536 env-sub-steps at 1 / 2 / 4 waves per SIMD — a second wave buys 1 %, four buy 9 %): a SIMD's measured VALU issue "
                        "rate grows slowly with occupancy — one wave64 instruction per 5.6 cycles for a lone dependent chain, per 4.2 with two waves, 3.0 with "
                        "four, 2.4 with eight, packed and unpacked alike (profiles/r03_a_valu_issue_probe.jsonl) — and this kernel, ONE wave per SIMD with "
                        "its own instruction-level parallelism, already issues one per 4.3 cycles: the rate two waves reach together, so a second wave has "
                        "nothing to add.  The issue model's 2-cycle peak needs >= 8 waves per SIMD, which 244-256 registers per lane rule out; of the rate "
                        "reachable at one or two waves per SIMD the kernel is at 0.98"}
        return {"bound": "valu", "achieved": ach / 1e9, "peak": simds * clock / 2 / 1e9, "unit": "G wave-instructions/s",
                "frac": ach / (simds * clock / 2), "frac_of_single_wave_issue": cnt["frac_of_single_wave_issue"], "fp32_lanes": lane,
                "source": "VALU wave-instructions per 65 536-env launch from " + src + "; duration = this run's HIP events; peak = one "
                          "wave64 VALU instruction per 2 cycles per SIMD at 2.4 GHz; one wave per SIMD can issue one per 4"}

    def timed_series(block, K):
        """The contract's timed region: EXACTLY K steps per block, barrier + synchronize on both sides of the series, wall
        time = max over ranks.  A K-step block is short (20 steps = 0.2 ms), so the block is repeated back to back —
        `repeats` times, until the series lasts MIN_SERIES_S — with a HIP event on the launch stream between consecutive
        blocks: `ms_per_step` = series wall time / (repeats x K), and the per-block event times give the median block
        (the roofline's duration) and the spread.  Returns a dict."""
        barrier()
        t0 = time.perf_counter()
        block(K)
        torch.cuda.synchronize(dev)
        est = max_over_ranks(time.perf_counter() - t0)           # the same decision on every rank
        reps = int(min(MAX_REPEATS, max(MIN_REPEATS, math.ceil(MIN_SERIES_S / max(est, 1e-7)))))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        barrier()
        t0 = time.perf_counter()
        evs[0].record(stream)            # torch's current stream == the stream the launches (or their fork / join) are ordered on
        for r in range(reps):
            block(K)
            evs[r + 1].record(stream)
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        bl = sorted(evs[r].elapsed_time(evs[r + 1]) for r in range(reps))
        return {"wall_s": wall, "repeats": reps, "block_s": wall / reps, "block_ev_ms": statistics.median(bl),
                "block_ev_ms_p10_p90": [bl[int(0.1 * (reps - 1))], bl[int(0.9 * (reps - 1))]],
                "block_ev_spread": (bl[int(0.9 * (reps - 1))] - bl[int(0.1 * (reps - 1))]) / statistics.median(bl)}

    class StepLeg:
        """nl envs of this rank (global ids from id_off) stepped by pnr_step / pnr_rollout into an obs ring, as `parts`
        handles over contiguous blocks of the env axis.  parts > 1: each handle's launches go to a stream of its own
        (forked from and joined to the launch stream per block), so launch t of one part does not wait for launch t of the
        other; graph=1: a K-step block is captured once in a hipGraph and replayed."""

        def __init__(self, nl, id_off, parts=1, graph=0, Tl=1, ring=32, n_act=16, sim=None, **engine_kw):
            self.nl, self.parts, self.graph, self.T = nl, parts, graph, Tl
            eng = dict(max_episode_steps=500, auto_reset=True, obs_layout=args.obs_layout, action_layout=args.action_layout)
            eng.update(engine_kw)
            if parts > 1 and (args.obs_layout != "env_major" or args.action_layout != "env_major"):
                raise SystemExit("--split needs env-major observations and actions (a part's rows are one contiguous span)")
            self.bounds = [shard_range(nl, parts, i) for i in range(parts)]
            self.envs = [PioneerVectorEnv(c, device=dev, seed=0, env_id_offset=id_off + o, simulation_config=sim,
                                          engine_config=EngineConfig(**eng)) for o, c in self.bounds]
            for e in self.envs:
                e.reset()
            e0 = self.envs[0]
            self.lib = e0.lib
            fm_o, fm_a = e0.feature_major_obs, e0.feature_major_act
            n_act = max(n_act, Tl)
            ring = max(ring, Tl)
            ring -= ring % Tl
            self.ring, self.n_act = ring, n_act
            self.acts = (torch.rand((n_act, 6, nl) if fm_a else (n_act, nl, 6), generator=g, device=dev) * 2 - 1) * \
                (amax[:, None] if fm_a else amax)
            self.obs = torch.empty((ring, 137, nl) if fm_o else (ring, nl, 137), dtype=torch.float32, device=dev)
            self.rew = torch.empty((ring, nl), dtype=torch.float32, device=dev)
            self.done = torch.empty((ring, nl), dtype=torch.uint8, device=dev)
            self.trunc = torch.empty((ring, nl), dtype=torch.uint8, device=dev)
            self.streams = [stream] if parts == 1 else [torch.cuda.Stream(dev) for _ in range(parts)]
            self._graphs = {}
            # per part: the argument tuples of one pass over the ring
            self.calls = []
            for o, c in self.bounds:
                if Tl == 1:
                    rows = [(V(self.acts[i % n_act, o:].data_ptr()) if not fm_a else V(self.acts[i % n_act].data_ptr()),
                             V(self.obs[i % ring, o:].data_ptr()) if not fm_o else V(self.obs[i % ring].data_ptr()),
                             V(self.rew[i % ring, o:].data_ptr()), V(self.done[i % ring, o:].data_ptr()),
                             V(self.trunc[i % ring, o:].data_ptr())) for i in range(math.lcm(n_act, ring))]
                else:
                    nslots = ring // Tl
                    # operand shapes must cover what one launch touches: Tl action slices, Tl output slices per slot
                    assert self.acts.shape[0] >= Tl and nslots >= 1 and nslots * Tl <= self.obs.shape[0] == self.rew.shape[0], \
                        (self.acts.shape, self.obs.shape, Tl)
                    rows = [(V(self.acts[0].data_ptr()), V(self.obs[sl * Tl].data_ptr()), V(self.rew[sl * Tl].data_ptr()),
                             V(self.done[sl * Tl].data_ptr()), V(self.trunc[sl * Tl].data_ptr())) for sl in range(nslots)]
                self.calls.append(rows)

        def _launch(self, K, sps):
            """K steps of every part, enqueued (launches interleaved over the parts)."""
            lib, Tl = self.lib, self.T
            m = len(self.calls[0])
            hs = [e._h for e in self.envs]
            for i in range(K // Tl):
                for pi in range(self.parts):
                    a, o, r, d, tr = self.calls[pi][i % m]
                    rc = lib.pnr_step(hs[pi], a, o, r, d, tr, None, sps[pi]) if Tl == 1 else \
                        lib.pnr_rollout(hs[pi], Tl, a, o, r, d, tr, sps[pi])
                    if rc:
                        _lib.check(rc, hs[pi])

        def _fork_launch_join(self, K, main):
            if self.parts == 1:
                self._launch(K, [V(main.cuda_stream)])
                return
            ev = torch.cuda.Event()
            ev.record(main)
            for st in self.streams:
                st.wait_event(ev)
            self._launch(K, [V(st.cuda_stream) for st in self.streams])
            for st in self.streams:
                e2 = torch.cuda.Event()
                e2.record(st)
                main.wait_event(e2)

        def block(self, K):
            """Enqueue exactly K steps, ordered after everything on the launch stream and before whatever follows on it."""
            assert K % self.T == 0
            if not self.graph:
                self._fork_launch_join(K, stream)
                return
            gr = self._graphs.get(K)
            if gr is None:
                torch.cuda.synchronize(dev)
                gr = torch.cuda.CUDAGraph()
                cap = torch.cuda.Stream(dev)
                with torch.cuda.stream(cap):
                    gr.capture_begin()
                    self._fork_launch_join(K, torch.cuda.current_stream(dev))
                    gr.capture_end()
                self._graphs[K] = gr
            gr.replay()

        def close(self):
            self._graphs.clear()
            for e in self.envs:
                e.close()

    def run_leg(leg, K, W):
        K -= K % leg.T
        W -= W % leg.T
        if W > 0:
            leg.block(W)
        return timed_series(leg.block, K), K, W

    split = args.split
    if split < 0:
        split = 1
    use_graph = args.graph if args.graph >= 0 else 0
    simc = SimulationConfig(gravity=args.gravity)
    main_leg = StepLeg(n, env_start, parts=split, graph=use_graph, Tl=T, ring=args.ring, sim=simc, mode=args.mode, randomize=args.randomize)
    ts, K, W = run_leg(main_leg, args.steps, args.warmup)
    elapsed, repeats = ts["block_s"], ts["repeats"]
    value = float(total_envs) * K / elapsed
    single = None
    if split > 1 or use_graph:
        # the same batch as ONE eager launch per step: the form whose kernel duration rocprofv3 can check directly
        main_leg.close()
        sl = StepLeg(n, env_start, parts=1, graph=0, Tl=T, ring=args.ring, sim=simc, mode=args.mode, randomize=args.randomize)
        s1, K1, _ = run_leg(sl, args.steps, args.warmup)
        lms = s1["block_ev_ms"] / (K1 // T)
        ab = n * (STEP_IO_BYTES * T + STATE_BYTES)
        single = {"value": float(total_envs) * K1 / s1["block_s"], "unit": "env-steps/s", "repeats": s1["repeats"], "ms_per_step": s1["block_s"] / K1 * 1e3,
                  "avg_launch_ms": lms, "achieved_GBps": ab / (lms * 1e-3) / 1e9, "frac": ab / (lms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                  "note": "one handle, one eager pnr_step launch per step: avg_launch_ms is the kernel duration a rocprofv3 --kernel-trace of "
                          "`bench.py --split 1` shows (profiles/)"}
        sl.close()
    else:
        main_leg.close()

    # the same batch as TWO handles of n/2 envs whose launches are independent (two streams, captured per
    # 32-step block in a hipGraph), meant to run launch t of one half under the dependent-launch boundary of the other
    split_leg = None
    if split == 1 and T == 1 and args.mode == "kinematic" and args.split_leg and n >= 2 * 4096 and \
            args.obs_layout == "env_major" and args.action_layout == "env_major":
        sl = StepLeg(n, env_start, parts=2, graph=1, Tl=1, ring=args.ring, sim=simc)
        s2, K2, _ = run_leg(sl, 32, 64)
        sl.close()
        ms2 = s2["block_ev_ms"] / K2
        ab = n * BYTES_PER_ENV_STEP
        split_leg = {"value": float(total_envs) * K2 / s2["block_s"], "unit": "env-steps/s", "handles_per_gpu": 2, "steps_per_graph_replay": K2,
                     "repeats": s2["repeats"], "ms_per_step": s2["block_s"] / K2 * 1e3, "avg_step_ms": ms2,
                     "achieved_GBps": ab / (ms2 * 1e-3) / 1e9, "frac": ab / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "note": "two handles of n/2 envs (global env ids: bit-identical outputs, test_split_batch_on_two_streams_equals_one_launch_"
                             "bit_for_bit), launches on two streams inside 32-step hipGraph replays; avg_step_ms = both launches of a step. "
                             "Not the headline: the gain over one launch per step is 0-7 % by graph size and box (DESIGN.md section 3)"}

    fused = None
    if T == 1 and args.fused_leg > 1 and args.mode == "kinematic":
        Tf = args.fused_leg
        fl = StepLeg(n, env_start, parts=1, graph=0, Tl=Tf, ring=max(args.ring, Tf), n_act=Tf, sim=simc)
        fs, fK, _ = run_leg(fl, max(Tf, args.steps), max(Tf, args.warmup))
        fl.close()
        fl_ms = fs["block_ev_ms"] / (fK // Tf)
        fb = n * (STEP_IO_BYTES * Tf + STATE_BYTES)
        fused = {"value": float(total_envs) * fK / fs["block_s"], "unit": "env-steps/s", "steps_per_launch": Tf, "steps": fK,
                 "repeats": fs["repeats"], "ms_per_step": fs["block_s"] / fK * 1e3, "avg_launch_ms": fl_ms,
                 "algorithmic_bytes_per_launch": fb, "achieved_GBps": fb / (fl_ms * 1e-3) / 1e9,
                 "frac": fb / (fl_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                 "note": "pnr_rollout: same kernel, T steps per launch with open-loop actions; state stays in registers"}

    def side_leg(nl, id_off, k, warm, note, bytes_per_env_step=BYTES_PER_ENV_STEP, sim=None, rollout_T=0, **engine_kw):
        """pnr_step on a separate env batch of nl envs per GPU: wall time is the max over ranks between
        barriers, avg_launch_ms is rank 0's HIP-event timing."""
        lring = 8 if nl >= 131072 else 32
        leg = StepLeg(nl, id_off, parts=1, graph=0, Tl=1, ring=lring, n_act=4, sim=sim, **engine_kw)
        st, k, _ = run_leg(leg, k, warm)
        lms = st["block_ev_ms"] / k
        roll = None
        if rollout_T > 1 and lring >= rollout_T:
            # the same steps as ONE pnr_rollout launch per rollout_T steps (open-loop actions)
            leg.close()
            leg = StepLeg(nl, id_off, parts=1, graph=0, Tl=rollout_T, ring=lring, n_act=rollout_T, sim=sim, **engine_kw)
            rs, rk, _ = run_leg(leg, max(4 * rollout_T, k - k % rollout_T), 3 * rollout_T)
            roll = {"steps_per_launch": rollout_T, "steps": rk, "repeats": rs["repeats"], "ms_per_step": rs["block_s"] / rk * 1e3,
                    "value": float(nl) * world * rk / rs["block_s"], "unit": "env-steps/s"}
        leg.close()
        if roll:
            note = note + "; `rollout`: pnr_rollout, the steps looped inside one launch"
        return {**({"rollout": roll} if roll else {}), "envs_per_gpu": nl, "steps": k, "repeats": st["repeats"],
                "avg_launch_ms": lms, "ms_per_step": st["block_s"] / k * 1e3,
                "value": float(nl) * world * k / st["block_s"], "unit": "env-steps/s",
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
        dynamic["roofline"] = dyn_valu_roofline(dynamic["avg_launch_ms"], dcnt)
        if "rollout" in dynamic:
            dynamic["rollout"]["value"] = float(total_envs) / (dynamic["rollout"]["ms_per_step"] * 1e-3)
    id_base += total_envs

    # weak scaling (the r01 headline): every rank steps its own 65 536-env shard
    weak = None
    if T == 1 and world > 1 and args.mode == "kinematic" and args.weak_envs > 0:
        weak = side_leg(args.weak_envs, id_base + rank * args.weak_envs, max(256, args.steps), 64,
                        f"{args.weak_envs} envs PER GPU on {world} ranks (weak scaling; not the metric's configuration)")
        weak["total_envs"] = args.weak_envs * world

    PRECISION_LABEL = {
        "f32": "float32-accurate products (two scaled fp16 planes per operand, 22 significant bits: measured at torch float32's own distance "
               "from float64) — the arithmetic of the reference's learner (float32 torch, pioneer_knm_train.py:47): the credited figure",
        True: "REDUCED-PRECISION VARIANT: bf16 MFMA operands (8 significant bits; gradients ~9e-3 relative from float32)",
        "bf16x3": "float32-accurate products as three bf16 planes per operand (24 bits, six MFMAs per product: r04's form)"}

    def ppo_leg(iters, global_mbs, precision=True, total=None):
        """BASELINE config[2] at N=1 (16 384 envs, full rollout + learn loop) and config[3] at N>1
        (65 536 envs in total sharded over the ranks, gradients all-reduced over RCCL/xGMI)."""
        from pioneer_amd.ppo import PPOConfig, PPOTrainer
        if total is None:
            total = args.ppo_envs if args.ppo_envs > 0 else (16384 if world == 1 else 65536)
        start, cnt = pdist.shard_range(total, world, rank)
        penv = PioneerVectorEnv(cnt, device=dev, seed=0, env_id_offset=start,
                                engine_config=EngineConfig(max_episode_steps=500, auto_reset=True, mode=args.mode))
        mbs = max(1, min(global_mbs // world, 32 * cnt))      # this rank's share of every global minibatch
        pcfg = PPOConfig(rollout_fragment_length=32, num_sgd_iter=4, sgd_minibatch_size=mbs, hip_kernels=precision)
        tr = PPOTrainer(penv, pcfg, use_graph=True)
        tr.train(); tr.train()                       # warm-up: eager iteration, then the graph-captured one
        barrier()
        tp = time.perf_counter()
        rs = [tr.train() for _ in range(iters)]
        barrier()
        tp = max_over_ranks(time.perf_counter() - tp)
        graphed = {"sampling": tr._graph is not None,
                   "sampler": ("pnr_ppo_rollout: the T-step closed loop (both nets, action draw, env step) as one resident launch"
                               if getattr(tr, "resident_rollout", False) else "pnr_mlp_act + pnr_step per step"),
                   "learner": ("hip kernels: pnr_mlp_train_step, 3 launches per update, no graph needed" if tr.learner.hip
                               else "torch autograd, eager"),
                   "learner_split_around_allreduce": bool(tr.learner.hip and world > 1)}
        finite = all(math.isfinite(float(r[k])) for r in rs for k in ("kl", "total_loss"))
        hip_learner = bool(tr.learner.hip)
        penv.close()
        del tr
        torch.cuda.empty_cache()
        steps = iters * 32 * total
        # ---- roofline of the learner (the kernels behind pnr_mlp_train_step): useful flops against the dense bf16 MFMA peak and the
        # bytes the three kernels move (FETCH_SIZE x 2 + WRITE_SIZE per update from a committed rocprofv3 --pmc pass over
        # tools/mlp_step_bench.py, quoted only while the learner sources are the ones that pass ran on) against the HBM peak
        updates = iters * 4 * ((32 * cnt) // mbs)
        learn_s = sum(r["learn_time_s"] for r in rs)
        us_per_update = learn_s / max(1, updates) * 1e6
        planes = pcfg.mlp_planes() if hip_learner else 0
        mac_per_sample_net = 106496 + 69632 + 110592           # forward, backward-data, weight gradients (csrc/pnr_mlp.h)
        flops = 2.0 * mac_per_sample_net * mbs * 2             # per update on this rank, both nets: the model's flops ..
        mfma_flops = flops * {0: 0, 1: 1, 2: 3, 3: 6}[planes]  # .. and what the matrix cores execute for them (plane pairs)
        roof = {"bound": None, "us_per_update": us_per_update,
                "minibatch_per_rank": mbs, "flops_per_update": flops, "mfma_flops_per_update": mfma_flops,
                "achieved_TFLOPs": flops / (us_per_update * 1e-6) / 1e12, "mfma_peak_TFLOPs": 2500.0,
                "mfma_frac": mfma_flops / (us_per_update * 1e-6) / 2.5e15,
                "timing": "learn_time_s of the timed iterations / updates (includes the epoch's gather and the record packing)"}
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", LEARNER_TRAFFIC)))
            have, want = doc.get("learner_sha16"), _lib.source_fingerprint(_lib.LEARNER_KERNEL_SOURCES)
            key = f"planes{planes}"
            if have != want:
                roof["bytes_per_update"] = None
                roof["bytes_source"] = f"stale: the learner kernels changed since the counter pass (profiles/{LEARNER_TRAFFIC}: {have}, this build: {want})"
            elif key not in doc or int(doc[key]["batch"]) != int(mbs):
                roof["bytes_per_update"] = None
                roof["bytes_source"] = f"profiles/{LEARNER_TRAFFIC} has no pass for {key} at {mbs} samples"
            else:
                b = float(doc[key]["bytes_per_update"])
                roof.update({"bytes_per_update": b, "hbm_GBps": b / (us_per_update * 1e-6) / 1e9, "hbm_peak_GBps": HBM_PEAK_GBPS,
                             "hbm_frac": b / (us_per_update * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                             "bytes_by_kernel": {k: v["bytes"] for k, v in doc[key]["kernels"].items()},
                             "bytes_source": f"profiles/{LEARNER_TRAFFIC} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on these sources, {want}; "
                                             "FETCH_SIZE doubled, Infinity-Cache hits counted; looked up, NOT measured in this run)"})
        except Exception as exc:
            roof["bytes_per_update"] = None
            roof["bytes_source"] = f"profiles/{LEARNER_TRAFFIC}: {type(exc).__name__}"
        # which roofline the update's arithmetic intensity puts it under (MFMA flops per counter byte against the ridge 2.5 PF / 8 TB/s)
        if roof.get("bytes_per_update"):
            inten = mfma_flops / roof["bytes_per_update"]
            roof["intensity_mfma_flop_per_byte"] = inten
            roof["bound"] = ("mfma" if inten > 2.5e15 / (HBM_PEAK_GBPS * 1e9) else "hbm")
            roof["bound_note"] = (f"{inten:.0f} MFMA flop per byte against a ridge of {2.5e15 / (HBM_PEAK_GBPS * 1e9):.0f}; neither roofline is approached "
                                  "(mfma_frac, hbm_frac): the kernels are bound by the dependent chain of a tile's phases, DESIGN.md section 3c")
        else:                                                   # no usable counter pass for this leg: no byte figure is quoted at all
            roof.pop("bytes_per_update", None)
            roof["bound"] = "mfma + latency (bytes not quoted: " + roof.get("bytes_source", "no counter pass") + ")"
        return {"value": steps / tp, "precision": PRECISION_LABEL.get(precision, str(precision)), "roofline": roof if hip_learner else None, "unit": "env-steps/s", "total_envs": total, "envs_per_gpu": cnt, "rollout_T": 32,
                "num_sgd_iter": 4, "sgd_minibatch_size": mbs * world, "sgd_minibatch_size_per_rank": mbs,
                "sgd_updates_per_iter": 4 * ((32 * cnt) // mbs), "mlp_dtype": pcfg.mlp_dtype(),
                "hip_graph": graphed, "iters": iters, "losses_finite": finite,
                "grad_allreduce": ({"backend": backend, "world_size": world,
                                    "bytes": 4 * (int(_lib.load_library().pnr_mlp_grad_floats()) if hip_learner else 205581),
                                    "per": ("minibatch; the two nets as two chains on two streams, each all-reducing its half of the bucket "
                                            "(the kernels' padded layout)" if hip_learner else "minibatch, one flat bucket")}
                                   if world > 1 else None),
                "sample_time_s": sum(r["sample_time_s"] for r in rs), "learn_time_s": sum(r["learn_time_s"] for r in rs),
                "note": "full loop: policy MLP 137-256-256 fwd per step, GAE, 4 SGD epochs, obs filter, grad all-reduce"}

    def emit(ppo):
        """ppo: the PPO legs that have a result so far (key -> dict; an error dict for the one that failed or was cut off)."""
        if rank != 0:
            return
        launches = K // T
        step_ms = ts["block_ev_ms"] / launches            # per pass over the rank's batch (all parts), median block, HIP events
        # per pass: T steps of action/obs/reward/flags traffic + ONE state read and write per env
        algo_bytes = n * (STEP_IO_BYTES * T + STATE_BYTES)
        achieved = algo_bytes / (step_ms * 1e-3) / 1e9
        traffic, traffic_source = looked_up("pmc_traffic.json", [f"{args.mode}:{args.obs_layout}:{n}:{T}", "hbm_bytes_per_launch"])
        kname = "pnr::step_kernel" if args.mode == "kinematic" else "pnr::dyn_step_kernel"
        variant = ("one pnr_step launch per step" if T == 1 else f"one pnr_rollout launch per {T} steps") if split == 1 else \
            (f"{split} handles of {n // split} envs (contiguous blocks, global env ids), their pnr_step launches independent of each other")
        variant += "; each timed block is one hipGraph replay" if use_graph else "; eager launches"
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
                       "steps_per_launch": T, "variant": variant, "handles_per_gpu": split, "hip_graph": bool(use_graph),
                       "obs_ring_slices": main_leg.ring, "parallelism": f"env-shard x{world}",
                       "timed_region": f"{repeats} back-to-back blocks of exactly {K} steps between barrier + synchronize; "
                                       "ms_per_step = wall / (repeats x steps)",
                       "series_wall_s": ts["wall_s"], "block_event_ms_median": ts["block_ev_ms"],
                       "block_event_ms_p10_p90": ts["block_ev_ms_p10_p90"], "block_event_spread_p10_p90": ts["block_ev_spread"]},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": algo_bytes // split, "launches_per_step": split,
                         "algorithmic_bytes_per_step": algo_bytes,
                         "avg_launch_ms": step_ms,
                         "timing": "HIP events on the launch stream between consecutive K-step blocks (rank 0; median block) / steps per block"
                                   + ("; the step's launches overlap, so avg_launch_ms is the time per STEP (all its launches) and the kernel "
                                      "durations of a rocprofv3 trace (which serialises dispatches) do not add up to it: `single_launch` is the "
                                      "same batch as one launch per step, whose duration the trace shows" if split > 1 else "")},
        }
        if single:
            out["roofline"]["single_launch"] = single
        if split_leg:
            out["split_streams"] = split_leg
        if fused:
            out["fused_rollout"] = fused
        if large:
            out["large_batch"] = large
        if dynamic:
            out["dynamics_randomized"] = dynamic
        if weak:
            out["weak_scaling"] = weak
        for key, res in ppo.items():
            if "." in key:                                    # "ppo_loop_65536.f32" -> out["ppo_loop_65536"]["f32"]
                top, sub = key.split(".", 1)
                out.setdefault(top, {"note": "BASELINE config[3] whole on ONE GPU: 65 536 envs, T = 32, four epochs of 32 768-sample minibatches — "
                                             "the N = 1 point of SURVEY 8(d)'s 1/2/4/8 table (at N > 1 `ppo_loop_f32` / `ppo_loop` are this workload sharded)"})[sub] = res
            else:
                out[key] = res
        lib = _lib.load_library()
        built, tree = lib.pnr_build_fingerprint().decode(), _lib.tree_fingerprint()
        out["build"] = {"binary": built, "tree": tree, "binary_is_this_tree": built == tree,
                        "env_kernel_sources_sha16": _lib.source_fingerprint(), "learner_kernel_sources_sha16": _lib.source_fingerprint(_lib.LEARNER_KERNEL_SOURCES),
                        "note": "binary / tree: per translation unit, sha256 over flags + sources + header (pnr_build_fingerprint, baked in at compile time; the "
                                "loader refuses a mismatch); the *_sources_sha16 are what the looked-up counter passes (roofline.traffic_source, "
                                "ppo_loop*.roofline.bytes_source) are checked against"}
        if not args.no_cpu_baseline and world == 1:      # contract: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    ppo = {}                 # key -> result of the PPO legs that have finished (insertion order = the order in the line)
    rc = 0
    if args.ppo_iters < 0:
        args.ppo_iters = 10 if args.mode == "kinematic" else 0      # ~150 ms of timed work per PPO leg
    if args.ppo_iters > 0:
        # The extra legs must never take the main result down: the line is still printed when one fails, but the process then ends
        # NON-ZERO — an exception is reported in place of ITS leg (rc 5), and a leg that does not come back (e.g. ranks out of step in
        # a collective) is cut off by a watchdog that prints the line with every FINISHED leg and the error in place of the one that
        # was running (rc 3).
        import threading
        it, half = args.ppo_iters, max(2, args.ppo_iters // 2)
        plan = [("ppo_loop_f32", lambda: ppo_leg(it, args.ppo_minibatch, "f32")),          # the credited loop: the reference learner's arithmetic
                ("ppo_loop", lambda: ppo_leg(it, args.ppo_minibatch, True))]                # its bf16 reduced-precision variant
        if world == 1 and args.ppo_65536 and not args.ppo_envs:
            plan += [("ppo_loop_65536.f32", lambda: ppo_leg(max(2, it // 3), args.ppo_minibatch, "f32", total=65536)),
                     ("ppo_loop_65536.bf16", lambda: ppo_leg(max(2, it // 3), args.ppo_minibatch, True, total=65536))]
        if world == 1 and args.ppo_f32:
            plan.append(("ppo_loop_bf16x3", lambda: ppo_leg(half, args.ppo_minibatch, "bf16x3")))
        if args.ppo_large_minibatch > 0 and args.ppo_large_minibatch != args.ppo_minibatch:
            plan.append(("ppo_loop_large_minibatch", lambda: ppo_leg(half, args.ppo_large_minibatch, True)))
        lock, state = threading.Lock(), {"done": False, "leg": None}

        def on_timeout():
            with lock:
                if state["done"]:
                    return
                state["done"] = True
                res = dict(ppo)
                if state["leg"]:
                    res[state["leg"]] = {"error": f"did not finish: the PPO legs' common budget of {args.ppo_timeout:.0f} s ran out in this leg"}
                emit(res)
                sys.stdout.flush()
                os._exit(3)
        wd = threading.Timer(args.ppo_timeout, on_timeout)
        wd.daemon = True
        wd.start()
        for key, fn in plan:
            with lock:
                state["leg"] = key
            try:
                res = fn()
            except Exception as exc:
                res = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                rc = 5
            with lock:
                ppo[key] = res
            if res.get("losses_finite") is False:
                rc = 5
            if "error" in res:
                break                                       # (a failed leg may have left a collective half done: no further legs)
        with lock:
            state["done"] = True
        wd.cancel()
    emit(ppo)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if os.environ.get("PNR_BENCH_DRYRUN") != "1" and os.environ.get("PNR_BENCH_SHARE_GPU") != "1":
            import torch                                   # device_count() only: nothing here touches the GPU
            ndev = torch.cuda.device_count()
            if ndev < args.gpus:
                print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible devices (found {ndev})", file=sys.stderr, flush=True)
                return 2
        return self_launch(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())

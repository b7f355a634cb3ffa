"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl"
on ROCm) or gloo on CPU.  Envs are sharded by contiguous blocks with NO data-path
collective (they never interact); the only exchanges are the PPO
gradient all-reduce, the obs-filter moments and scalar metrics.
"""
import os
from typing import Iterable, Tuple

import torch
import torch.distributed as dist


_SOLO = False


class solo:
    """Inside this context the process behaves as a ONE-rank job although a process group exists: `is_dist()` is False (no
    collective is issued by anything that asks it first), `world_info()` reports rank 0 of 1 (with the real local rank: the GPU).
    What `launch.train(trial_parallel=True)` wraps a rank's own trials in — the reference's scaling axis (Tune runs
    `num_samples` independent trials, pioneer_knm_train.py:43-44): no traffic between the GPUs at all while they train."""

    def __enter__(self):
        global _SOLO
        self._prev, _SOLO = _SOLO, True
        return self

    def __exit__(self, *exc):
        global _SOLO
        _SOLO = self._prev
        return False


def world_info() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    if _SOLO:
        return 0, int(os.environ.get("LOCAL_RANK", "0")), 1
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend: str = None, device: torch.device = None) -> Tuple[int, int]:
    """Initialise the default process group if WORLD_SIZE > 1.  Returns (rank, world)."""
    rank, _, world = world_info()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # PNR_DIST_BACKEND=gloo: rehearsal of the N>1 path with several ranks sharing one GPU
            # (RCCL refuses two ranks on one device)
            backend = os.environ.get("PNR_DIST_BACKEND") or \
                ("nccl" if (device is not None and device.type == "cuda") else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


_FORCE = False


def force_collectives(on: bool = True) -> None:
    """Take the several-rank code paths (gradient bucket -> all-reduce -> pnr_mlp_adam, the two-chain learner, the filter / metric
    all-reduces) with whatever process group is initialised, a ONE-rank group included.  What a one-GPU box can still check of the
    N > 1 path: the collectives then are real RCCL calls with RCCL's asynchronous stream semantics (a world-size-1 all-reduce is the
    identity on the data, so results must equal the no-collective run bit for bit).  tests/test_gpu_multirank.py."""
    global _FORCE
    _FORCE = bool(on)


def is_dist() -> bool:
    if (not _SOLO) and _FORCE and dist.is_available() and dist.is_initialized():
        return True
    return (not _SOLO) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def run_trials(num_samples: int, run_one, trial_parallel: bool = False):
    """`run_one(trial_index) -> row` for trials 0 .. num_samples - 1.  Default: every rank runs every trial together (each
    trial is data-parallel over the ranks).  trial_parallel with several ranks: rank r runs trials r, r + world, ... on its own
    (`solo`), and the rows are gathered once at the end — every rank returns all rows in trial order."""
    rank, _, world = world_info()
    if not (trial_parallel and is_dist()):
        return [run_one(t) for t in range(num_samples)]
    # The ranks exchange nothing while they train, so the gather at the end waits for the SLOWEST rank — a whole trial longer than
    # the fastest when num_samples % world != 0, or however far the trial times drift apart.  The default group's collective
    # timeout (10 minutes on nccl / RCCL, whose watchdog then aborts the job with no result rows) is the wrong clock for that: the
    # rows travel through a gloo group of their own (host objects anyway) whose timeout is days.  Created here, while the ranks
    # are still together.
    import datetime
    rows_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(days=7))
    with solo():
        mine = [(t, run_one(t)) for t in range(rank, num_samples, world)]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine, group=rows_group)
    dist.destroy_process_group(rows_group)
    return [row for _, row in sorted((x for part in gathered for x in part), key=lambda tr: tr[0])]


def broadcast_object(obj, src: int = 0):
    """The same Python object on every rank (rank `src`'s)."""
    if not is_dist():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def shard_range(total_envs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [start, start + count) of the global env axis owned by `rank`.
    The remainder goes to the lowest ranks, so sizes differ by at most one."""
    if not (0 <= rank < world):
        raise AssertionError(f"rank {rank} outside world {world}")
    base, rem = divmod(int(total_envs), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def allreduce_mean_grads(params: Iterable[torch.nn.Parameter]) -> None:
    """Data-parallel gradient averaging as ONE flat bucket (0.82 MB for the 205 581-parameter
    PPO nets): a single small all-reduce is latency-bound on xGMI, so bucketing by layer would
    only multiply the latency."""
    if not is_dist():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def allreduce_mean_(t: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks of an already-flat bucket."""
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(dist.get_world_size())
    return t


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_max_(t: torch.Tensor) -> torch.Tensor:
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def allreduce_min_(t: torch.Tensor) -> torch.Tensor:
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t


def broadcast_module_(module: torch.nn.Module, src: int = 0) -> None:
    """Make every rank start from rank `src`'s weights."""
    if not is_dist():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def barrier() -> None:
    if is_dist():
        dist.barrier()

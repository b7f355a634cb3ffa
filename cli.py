#!/usr/bin/env python3
"""Command line for the PPO driver.

Keeps the reference CLI's surface (cli.py:12-40 there: sub-command ``pioneer-train-kinem`` with
``-e/--experiment``, ``-c/--checkpoint-freq``, ``-n/--num-samples``, ``-w/--num-workers``,
``--no-monitor``, and ``tensorboard -e EXPERIMENT``, cli.py:43-55; ``tracking.training_root`` and a ``logging`` dictConfig read
from ``config.yaml`` next to this file, as the reference's config.yaml:1-24) on top of ``pioneer_amd.launch.train``, plus
``pioneer-eval`` (the role of the reference's temp/pioneer_eval.py: restore a checkpoint, roll out, record) and the engine's own
switches (``--restore CHECKPOINT``, ``--trial-parallel``, ``--mode``, ``--precision``).  For several GPUs run it under
``python -m torch.distributed.run --nproc-per-node N cli.py pioneer-train-kinem ...`` (``--trial-parallel``: one trial per GPU
at a time, the reference's own parallelism; default: every trial data-parallel over the GPUs).
"""
import argparse
import logging
import logging.config
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULTS = {"tracking": {"training_root": os.path.join(HERE, "pioneer_runs")}, "logging": None}


def load_settings() -> dict:
    settings = {k: (dict(v) if isinstance(v, dict) else v) for k, v in DEFAULTS.items()}
    path = os.path.join(HERE, "config.yaml")
    if os.path.exists(path):
        import yaml
        with open(path) as fh:
            user = yaml.safe_load(fh) or {}
        settings["tracking"].update(user.get("tracking") or {})
        settings["logging"] = user.get("logging")
    if settings["logging"]:
        logging.config.dictConfig(settings["logging"])
    else:
        logging.basicConfig(level=logging.INFO, stream=sys.stderr,
                            format="%(asctime)s %(levelname)-8s [%(name)s] - %(message)s")
    return settings


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="cli.py")
    sub = ap.add_subparsers(dest="command", required=True)
    tr = sub.add_parser("pioneer-train-kinem", help="PPO on the Pioneer kinematic env (HIP engine)")
    tr.add_argument("-e", "--experiment", required=True, help="experiment name")
    tr.add_argument("-c", "--checkpoint-freq", type=int, default=10)
    tr.add_argument("-n", "--num-samples", type=int, default=128, help="number of trials")
    tr.add_argument("-w", "--num-workers", type=int, default=1, help="scales the env batch: workers x envs-per-worker")
    tr.add_argument("--no-monitor", action="store_true")
    tr.add_argument("--iterations", type=int, default=1000)
    tr.add_argument("--envs-per-worker", type=int, default=4096)
    tr.add_argument("--mode", choices=["kinematic", "dynamic"], default="kinematic")
    tr.add_argument("--precision", choices=["f32", "bf16", "bf16x3", "torch"], default="f32",
                    help="the learner's arithmetic on the hand-written kernels: 'f32' (default) float32-accurate products — two scaled fp16 planes per "
                         "operand — the arithmetic of the reference's float32 torch learner (pioneer_knm_train.py:47); 'bf16' the reduced-precision "
                         "fast variant (8 significant bits per operand, ~2.3x the env-steps/s); 'bf16x3' three bf16 planes per operand; 'torch' the "
                         "float32 torch formulation itself")
    tr.add_argument("--restore", default=None, metavar="CHECKPOINT", help="start every trial from this PPOTrainer.save() file")
    tr.add_argument("--trial-parallel", action="store_true",
                    help="several GPUs: rank r runs trials r, r + world, ... on its own (no traffic between the GPUs)")
    tb = sub.add_parser("tensorboard", help="serve the experiment's event files with TensorBoard (reference cli.py:43-55)")
    tb.add_argument("-e", "--experiment", required=True, help="experiment name")
    tb.add_argument("--port", type=int, default=6006)
    ev = sub.add_parser("pioneer-eval", help="roll a saved policy out in the single-env facade, optionally record a GIF")
    ev.add_argument("-k", "--checkpoint", required=True, help="a checkpoint_*.pt written by pioneer-train-kinem")
    ev.add_argument("--episodes", type=int, default=3)
    ev.add_argument("--max-steps", type=int, default=500)
    ev.add_argument("--gif", default=None)
    ev.add_argument("--stochastic", action="store_true")
    ev.add_argument("--mode", choices=["kinematic", "dynamic"], default="kinematic")
    return ap


def launch_tensorboard(tensorboard_root: str, port: int = 6006) -> int:
    """pioneer/util.py:35-42 of the reference starts TensorBoard in-process and waits for Enter.  TensorBoard is a third-party
    package that this image does not carry: when it is importable it is started as a CHILD process on the experiment's event
    files (written by pioneer_amd/tb.py) until Enter / EOF; otherwise the command says where the files are and fails."""
    import importlib.util
    import subprocess
    n_events = sum(f.startswith("events.out.tfevents.") for _, _, fs in os.walk(tensorboard_root) for f in fs)
    if importlib.util.find_spec("tensorboard") is None:
        print(f"tensorboard is not installed here; {n_events} event file(s) under {tensorboard_root} "
              f"(serve them with: tensorboard --bind_all --port {port} --logdir {tensorboard_root})", file=sys.stderr)
        return 3
    proc = subprocess.Popen([sys.executable, "-m", "tensorboard.main", "--bind_all", "--port", str(port), "--logdir", tensorboard_root])
    logging.getLogger(__name__).info("Launched TensorBoard on port %d for %s (%d event files)", port, tensorboard_root, n_events)
    try:
        input("\nPress Enter to exit (this will terminate TensorBoard)\n")
    except EOFError:
        pass
    proc.terminate()
    return 0


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    settings = load_settings()
    if args.command == "tensorboard":
        return launch_tensorboard(os.path.join(settings["tracking"]["training_root"], args.experiment), args.port)
    if args.command == "pioneer-eval":
        import json
        from pioneer_amd.evaluate import evaluate
        eng = None
        if args.mode == "dynamic":        # the launcher's dynamics-mode default: the inertia-scaled motor
            from pioneer_amd import EngineConfig
            eng = EngineConfig(mode="dynamic", pd_kp=400.0, pd_kd=40.0, pd_inertia_scaled=True)
        res = evaluate(args.checkpoint, args.episodes, args.max_steps, args.gif, mode=args.mode,
                       deterministic=not args.stochastic, frame_stride=2, engine_config=eng)
        print(json.dumps(res))
        return 0
    from pioneer_amd.launch import RESULT_COLUMNS, dump, train
    out_dir = os.path.join(settings["tracking"]["training_root"], args.experiment)
    rows = train(results_dir=out_dir, checkpoint_freq=args.checkpoint_freq, num_samples=args.num_samples,
                 num_workers=args.num_workers, monitor=not args.no_monitor,
                 training_iterations=args.iterations, envs_per_worker=args.envs_per_worker, mode=args.mode,
                 restore=args.restore, trial_parallel=args.trial_parallel,
                 hip_kernels=False if args.precision == "torch" else args.precision)
    if int(os.environ.get("RANK", "0")) == 0:
        print("Results:\n\n" + dump(rows, RESULT_COLUMNS) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())

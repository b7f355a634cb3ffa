"""cli.py: the reference CLI's argument surface (cli.py:12-40 there) on CPU; the run itself on the GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_parser_keeps_the_reference_options():
    import cli
    ap = cli.build_parser()
    a = ap.parse_args(["pioneer-train-kinem", "-e", "exp1"])
    assert (a.experiment, a.checkpoint_freq, a.num_samples, a.num_workers, a.no_monitor) == ("exp1", 10, 128, 1, False)
    a = ap.parse_args(["pioneer-train-kinem", "--experiment", "x", "-c", "5", "-n", "2", "-w", "3", "--no-monitor"])
    assert (a.checkpoint_freq, a.num_samples, a.num_workers, a.no_monitor) == (5, 2, 3, True)
    with pytest.raises(SystemExit):
        ap.parse_args(["pioneer-train-kinem"])                      # -e is required, as in the reference
    e = ap.parse_args(["pioneer-eval", "-k", "ck.pt", "--gif", "o.gif"])
    assert e.checkpoint == "ck.pt" and e.gif == "o.gif" and e.episodes == 3
    a = ap.parse_args(["pioneer-train-kinem", "-e", "x", "--restore", "ck.pt", "--trial-parallel"])
    assert a.restore == "ck.pt" and a.trial_parallel
    assert a.precision == "f32" and ap.parse_args(["pioneer-train-kinem", "-e", "x", "--precision", "bf16"]).precision == "bf16"
    with pytest.raises(SystemExit):
        ap.parse_args(["pioneer-train-kinem", "-e", "x", "--precision", "fp8"])
    t = ap.parse_args(["tensorboard", "-e", "exp1"])                # the reference's second sub-command (cli.py:43-55)
    assert t.command == "tensorboard" and t.experiment == "exp1" and t.port == 6006


def test_shipped_config_yaml_has_the_reference_keys():
    import cli
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config.yaml")))
    assert set(cfg) == {"tracking", "logging"} and "training_root" in cfg["tracking"]
    assert cfg["logging"]["version"] == 1 and cfg["logging"]["root"]["handlers"] == ["console_handler"]
    s = cli.load_settings()                                          # the dictConfig is valid
    assert s["tracking"]["training_root"] == cfg["tracking"]["training_root"]


def test_tensorboard_command_without_the_package(tmp_path, monkeypatch, capsys):
    """No TensorBoard in this image: the command names the event files and fails (exit code 3) instead of pretending."""
    import importlib.util
    import cli
    if importlib.util.find_spec("tensorboard") is not None:
        pytest.skip("tensorboard is installed")
    monkeypatch.setitem(cli.DEFAULTS["tracking"], "training_root", str(tmp_path))
    monkeypatch.setattr(cli, "HERE", str(tmp_path))                  # no config.yaml: the defaults
    d = tmp_path / "exp1" / "PPO_Pioneer-v1_00000"
    d.mkdir(parents=True)
    (d / "events.out.tfevents.1.host").write_bytes(b"")
    assert cli.main(["tensorboard", "-e", "exp1"]) == 3
    err = capsys.readouterr().err
    assert "1 event file(s)" in err and "tensorboard --bind_all" in err


def test_settings_default_and_yaml_override(tmp_path, monkeypatch):
    import cli
    monkeypatch.setattr(cli, "HERE", str(tmp_path))
    s = cli.load_settings()
    assert s["tracking"]["training_root"].endswith("pioneer_runs") and s["logging"] is None
    (tmp_path / "config.yaml").write_text("tracking:\n  training_root: /tmp/somewhere\n")
    assert cli.load_settings()["tracking"]["training_root"] == "/tmp/somewhere"


@pytest.mark.gpu
def test_train_then_eval_through_the_cli(tmp_path, monkeypatch, capsys):
    import cli
    monkeypatch.setitem(cli.DEFAULTS["tracking"], "training_root", str(tmp_path))
    monkeypatch.setattr(cli, "HERE", str(tmp_path))                  # not the repo's config.yaml: the defaults above
    rc = cli.main(["pioneer-train-kinem", "-e", "smoke", "-c", "1", "-n", "1", "-w", "1", "--no-monitor",
                   "--iterations", "2", "--envs-per-worker", "256"])
    assert rc == 0
    assert not list((tmp_path / "smoke" / "PPO_Pioneer-v1_00000").glob("monitor_*.gif"))
    out = capsys.readouterr().out
    assert "episode_reward_mean" in out and "00000" in out          # the github-style results table
    tdir = tmp_path / "smoke" / "PPO_Pioneer-v1_00000"
    assert (tdir / "checkpoint_final.pt").exists()
    rc = cli.main(["pioneer-eval", "-k", str(tdir / "checkpoint_final.pt"), "--episodes", "1", "--max-steps", "6",
                   "--gif", str(tmp_path / "e.gif")])
    assert rc == 0 and (tmp_path / "e.gif").exists() and "episode_rewards" in capsys.readouterr().out
    # --restore: a second experiment continues from the first one's checkpoint (iteration and timestep counters carry on)
    rc = cli.main(["pioneer-train-kinem", "-e", "resumed", "-c", "0", "-n", "1", "--no-monitor", "--iterations", "1",
                   "--envs-per-worker", "256", "--restore", str(tdir / "checkpoint_final.pt")])
    assert rc == 0
    import json
    row = json.loads((tmp_path / "resumed" / "PPO_Pioneer-v1_00000" / "result.json").read_text().strip().splitlines()[-1])
    assert row["training_iteration"] == 3 and row["timesteps_total"] == 3 * 32 * 256
    # --precision f32: the same run with float32-accurate kernels (params.json records it); a bf16 checkpoint restores into it
    # (same master weights and Adam state: the operand precision is not part of a checkpoint)
    rc = cli.main(["pioneer-train-kinem", "-e", "f32run", "-c", "0", "-n", "1", "--no-monitor", "--iterations", "2",
                   "--envs-per-worker", "256", "--precision", "f32", "--restore", str(tdir / "checkpoint_final.pt")])
    assert rc == 0
    params = json.loads((tmp_path / "f32run" / "PPO_Pioneer-v1_00000" / "params.json").read_text())
    assert params["hip_kernels"] == "f32"
    row = json.loads((tmp_path / "f32run" / "PPO_Pioneer-v1_00000" / "result.json").read_text().strip().splitlines()[-1])
    assert row["training_iteration"] == 4 and all(row[k] == row[k] for k in ("kl", "total_loss"))      # 2 restored + 2


@pytest.mark.gpu
def test_monitor_records_an_episode_per_checkpoint(tmp_path):
    """monitor=True (RLlib's episode recording, pioneer_knm_train.py:51): a GIF next to every checkpoint."""
    from PIL import Image
    from pioneer_amd.launch import train
    from pioneer_amd.ppo import PPOConfig
    train(results_dir=str(tmp_path), checkpoint_freq=1, num_samples=1, num_workers=1, monitor=True, training_iterations=2,
          envs_per_worker=256, ppo_config=PPOConfig(rollout_fragment_length=8, num_sgd_iter=1, sgd_minibatch_size=2048),
          monitor_steps=9)
    d = tmp_path / "PPO_Pioneer-v1_00000"
    gifs = sorted(p.name for p in d.glob("monitor_*.gif"))
    assert gifs == ["monitor_1.gif", "monitor_2.gif", "monitor_final.gif"]
    assert Image.open(d / "monitor_final.gif").n_frames >= 1

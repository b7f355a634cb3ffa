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

#!/usr/bin/env python3
"""CLI mirroring the reference's cli.py:12-40 (`pioneer-train-kinem`): same options, same
config.yaml keys, same result table; the training itself is pioneer_amd.launch.train (PPO on the
HIP env, one process per GPU; launch under torch.distributed.run for several GPUs)."""
import logging.config
import os

import click
import yaml

from pioneer_amd.launch import RESULT_COLUMNS, dump, train


@click.command(name='pioneer-train-kinem')
@click.option('-e', '--experiment', 'experiment', required=True, type=str, help='experiment name')
@click.option('-c', '--checkpoint-freq', 'checkpoint_freq', default=10, type=int, help='checkpoint frequency (default: 10)')
@click.option('-n', '--num-samples', 'num_samples', default=128, type=int, help='number of search samples (default: 128)')
@click.option('-w', '--num-workers', 'num_workers', default=1, type=int, help='number of rollout workers (default: 1)')
@click.option('--no-monitor', 'no_monitor', is_flag=True, help='disable monitoring')
@click.option('--iterations', 'iterations', default=1000, type=int, help='training iterations per trial (reference: 1000)')
@click.option('--envs-per-worker', 'envs_per_worker', default=4096, type=int, help='device-resident envs per worker')
@click.option('--mode', 'mode', default='kinematic', type=click.Choice(['kinematic', 'dynamic']))
def cli_pioneer_train_kinem(experiment, checkpoint_freq, num_samples, num_workers, no_monitor, iterations,
                            envs_per_worker, mode):
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'config.yaml'), 'r') as config_file:
        config = yaml.safe_load(config_file)
        logging.config.dictConfig(config['logging'])
    experiment_dir = os.path.join(config['tracking']['training_root'], experiment)
    df = train(results_dir=experiment_dir, checkpoint_freq=checkpoint_freq, num_samples=num_samples,
               num_workers=num_workers, monitor=not no_monitor, training_iterations=iterations,
               envs_per_worker=envs_per_worker, mode=mode)
    if int(os.environ.get('RANK', '0')) == 0:
        print(f'Results: \n\n{dump(df, RESULT_COLUMNS)}\n\n\n')


@click.group()
def cli():
    pass


cli.add_command(cli_pioneer_train_kinem)

if __name__ == '__main__':
    cli()

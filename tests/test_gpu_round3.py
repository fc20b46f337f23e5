"""Round-3 GPU tests (VERDICT r02 "Next" #1, #2, #5), all through the C ABI / the shipped entry points:

* ``bench.py --gpus N`` starts its N ranks itself and the JSON line carries the executed-work roofline fraction and both floors;
* parity at the size and length of the workload: a 20-step, 300-residue trajectory with unscaled injected noise against
  oracle.sampler_ref, the same trajectory with the receptive-field pruning on vs off, bit-compared in deterministic mode."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import score_model_ref as smr
from oracle import sampler_ref as spr
from helpers import elem_err, rel_err, to_graph

pytestmark = pytest.mark.gpu
T = torch.from_numpy
CFG = smr.ScoreModelConfig(latent_vocab=64)
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    from disco_diffdock_amd import build
    build.build(verbose=False)
    return torch.device('cuda:0')


def _bench(*args):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus2_launches_two_ranks_by_itself(dev):
    """``python bench.py --gpus 2`` with no torchrun in the command runs two ranks (here: both on this box's one GPU over gloo)
    and rank 0 prints ONE line with n_gpus = 2; --gpus beyond the visible devices without --single-device fails loudly."""
    out = _bench('--gpus', '2', '--backend', 'gloo', '--single-device', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-alt',
                 '--no-extras', '--no-device-loop')
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['value'] > 0
    if torch.cuda.device_count() < 8:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'], capture_output=True, text=True,
                           env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}, timeout=300)
        assert r.returncode != 0 and 'GPU(s) are visible' in r.stderr


def test_bench_line_reports_executed_work_and_both_floors(dev):
    """roofline.frac is the EXECUTED fraction (<= 1, = flop_per_launch / avg_launch_ms / peak), the pruning-off floor and the
    pocket-bound workload are in the line, and the per-step executed-edge fractions cover the 20 steps."""
    out = _bench('--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-alt', '--no-device-loop')
    rf = out['roofline']
    assert 0.0 < rf['frac'] <= 1.0
    assert abs(rf['frac'] - rf['flop_per_launch'] / (rf['avg_launch_ms'] * 1e-3) / 1e12 / rf['peak']) < 1e-6 * rf['frac'] + 1e-9
    assert abs(rf['achieved'] - rf['frac'] * rf['peak']) < 1e-6 * rf['achieved']
    ex = out['extra']
    assert ex['pruning_off']['edges_executed_over_unpruned'] == pytest.approx(1.0)
    assert ex['pruning_off']['value'] > 0 and ex['pocket_bound']['value'] > 0
    assert ex['pocket_bound']['min_cross_edges_per_sample_over_steps'] > 0          # the pocket-bound samples never lose contact
    assert ex['pocket_bound']['edges_executed_over_unpruned'] >= rf['edges_executed_over_unpruned'] - 1e-9
    assert len(ex['per_step']) == 20 and all(0 < s['edges_executed_over_unpruned'] <= 1 for s in ex['per_step'])

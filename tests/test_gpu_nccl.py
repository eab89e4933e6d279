"""GPU tests that need a FRESH process (the pytest process has initialised the GPU; a child that initialises its own HIP
context is an ordinary subprocess, never a re-exec):
  * RCCL: `MetaTrainer.meta_iteration` through the real all-reduce with backend nccl at world_size 1 -- the collective's
    initialisation, device binding and stream ordering execute on hardware (`src/util/meta_run.py:237-243`,
    `src/train_meta.py:361-373` are what it replaces); N > 1 differs only in the ring, which one GPU cannot run;
  * BASELINE configs[4]: `python -m eosvos_amd.train_meta with YouTube-VOS ...` with the concurrent validation process
    (`src/train_meta.py:132-201`, `src/util/evaluate.py:34-40,361-382`) on the same GPU.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def clean_env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE'):
        env.pop(k, None)
    env.update(extra)
    return env


def test_meta_iteration_through_rccl_world_size_1(tmp_path):
    out = str(tmp_path / 'nccl.pt')
    p = subprocess.run([sys.executable, os.path.join(HERE, 'mp_workers', 'nccl_ws1_worker.py'), out], env=clean_env(),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = torch.load(out, weights_only=False)
    assert r['backend'] == 'nccl'
    assert r['all_reduce_calls'] == 4                      # 2 meta-iterations x {one task per rank, two in flight}
    assert r['mode_flag_calls'] == 2                       # + the matrix-mode verdict, once per loaded state (MetaTrainer._collective_mode_check)
    for tag in ('one', 'two'):
        # the sum over ONE rank is the identity: bit-identical to the trainer without a process group, i.e. the
        # all-reduce is ordered after the tasks' gradient accumulation and before the outer step on every stream
        assert r[tag]['equal'] and r[tag]['finite'] and r[tag]['step'] == 2, r[tag]
        assert r[tag]['losses'] == r[tag]['losses_ref']
        assert r[tag]['moved'] > 0


def test_bench_self_launch_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n + 1), '--steps', '1'], env=clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and 'GPU(s) visible' in p.stderr and p.stdout.strip() == ''


def test_concurrent_validation_process_beside_meta_training(tmp_path):
    """configs[4] on one GPU at 96x160: the trainer (a fresh `python -m eosvos_amd.train_meta` process) starts the
    validation child before it touches the GPU, replaces the snapshot every `vis_interval` meta-iterations, the child
    evaluates snapshots on its own engine WHILE meta-iterations run, writes last_* / best_* checkpoints and prediction
    PNGs, and is gone when the trainer exits."""
    save_dir = str(tmp_path / 'models')
    env = clean_env(EOSVOS_SYNTHETIC_SIZE='96x160', EOSVOS_SYNTHETIC_FRAMES='4', EOSVOS_NUM_META_ITERS='6')
    cmd = [sys.executable, '-m', 'eosvos_amd.train_meta', 'with', 'YouTube-VOS', 'meta_batch_size=2', 'num_epochs.train=2',
           'num_epochs.eval=3', 'vis_interval=1', f'save_dir={save_dir}', 'env_suffix=c4']              # (no `data/` under the repo: synthetic sequences)
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    run = os.path.join(save_dir, 'c4')
    iters = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{') and '"mode": "meta"' in l]
    assert [i['meta_iter'] for i in iters] == [1, 2, 3, 4, 5, 6]
    assert all(len(i['meta_losses']) == 2 and all(l == l for l in i['meta_losses']) for i in iters)        # no NaN
    ev = [json.loads(l) for l in open(os.path.join(run, 'eval_log.jsonl'))]
    assert len(ev) >= 1 and ev[-1]['meta_iter'] == 6                   # the final snapshot is always evaluated
    # at least one snapshot was evaluated while >= 2 later meta-iterations ran: an evaluated meta_iter below the last
    # (the child picks up whatever snapshot is current when it finishes the previous one)
    assert len(ev) >= 2 and ev[0]['meta_iter'] <= 4, ev
    key = ev[-1]['dataset']
    assert os.path.exists(os.path.join(run, f'last_{key}_meta_iter.model'))
    assert os.path.exists(os.path.join(run, f'best_{key}_meta_iter.model')) == (max(e['mean_J'] for e in ev) > 0)
    assert os.path.exists(os.path.join(run, 'last_meta_iter.model'))
    pngs = []
    for d, _, files in os.walk(os.path.join(run, 'best_eval_preds')):
        pngs += [f for f in files if f.endswith('.png')]
    assert len(pngs) >= 4
    # child gone: its pid was logged by nobody, so check there is no process left with our run dir on its command line
    left = subprocess.run(['pgrep', '-f', run], capture_output=True, text=True).stdout.split()
    assert not left, left


def test_c_abi_rccl_allreduce_world_size_1(tmp_path):
    """`eosvos_comm_unique_id / _init_rank / _destroy` + `eosvos_allreduce_sum` (`allreduce_sum(flat, n, comm)`):
    the library's own RCCL collective, no torch.distributed in the process, in a fresh process on the GPU.  Also: destroying
    the engine whose learned state the others alias FIRST leaves them with a valid copy (`eosvos_unalias_state` semantics,
    ADVICE r04)."""
    out = str(tmp_path / 'rccl_cabi.pt')
    env = clean_env()
    p = subprocess.run([sys.executable, os.path.join(HERE, 'mp_workers', 'rccl_cabi_worker.py'), out], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = torch.load(out, weights_only=False)
    assert r['identity']
    for tag in ('one', 'two'):
        assert r[tag]['equal'] and r[tag]['finite'] and r[tag]['moved'] > 0, r[tag]
        assert r[tag]['losses'] == r[tag]['losses_ref']

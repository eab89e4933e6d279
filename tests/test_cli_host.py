"""CPU checks of the entry points' host logic (stand-in engine of tests/fake_engine.py): the parent checkpoint named by
`parent_model.<key>.paths` reaches the engine (`src/train_meta.py:91-96`, `src/util/evaluate.py:46-50`), the meta-train
loop runs until it is stopped with `vis_interval` checkpoints (`:203-286`), `resume_meta_run_epoch_mode` LAST / BEST_<KEY>
(`:70-77`), the validation child never outlives a failed trainer, and the per-object train frame of YouTube-VOS objects
that first appear after frame 0 (`evaluate.py:132-137`, `data/youtube.py:131-143`)."""
import json
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'mp_workers'))


@pytest.fixture()
def fake_cli(monkeypatch):
    """train_meta / eval_worker with the stand-in model class but the PRODUCT's parent-state loading."""
    import common
    from eosvos_amd import helper_func, train_meta
    import fake_engine

    def init(architecture='DeepLabV3Plus', encoder='resnet50', batch_norm=None, **kw):
        monkeypatch.setattr(helper_func, 'DeepLabV3Plus', fake_engine.FakeDeepLab)
        return helper_func.init_parent_model(architecture, encoder, kw.pop('train_encoder', True), batch_norm=batch_norm, **kw)
    monkeypatch.setattr(train_meta, 'init_parent_model', init)
    fake_engine.FakeEngine.instances.clear()
    return train_meta, fake_engine, common


def _parent_file(path, shift):
    from eosvos_amd import synthetic
    sd = synthetic.synthetic_state('resnet50')
    sd['backbone.bn1.running_mean'] = sd['backbone.bn1.running_mean'] + shift
    sd['backbone.conv1.weight'] = sd['backbone.conv1.weight'] * 0 + shift
    torch.save(sd, path)
    return sd


def test_parent_checkpoint_reaches_the_engine(tmp_path, fake_cli, capsys):
    train_meta, fake_engine, common = fake_cli
    sd = _parent_file(tmp_path / 'parent.model', 0.75)
    mt = train_meta.main(['with', 'YouTube-VOS', 'meta_batch_size=1', 'num_epochs.train=1', f'save_dir={tmp_path}', 'env_suffix=p',
                          f'parent_model.train.paths=[{tmp_path / "parent.model"}]'], height=common.H, width=common.W,
                         num_meta_iters=1, data_root=str(tmp_path / 'none'), eval_cmd=False, device='cpu')
    eng = fake_engine.FakeEngine.instances[0]
    gamma, beta, mean, var = eng.norm_args
    assert torch.equal(mean[:64], sd['backbone.bn1.running_mean'])          # the file's statistics, not the synthetic ones
    assert float(mt.state[mt.n_lr]) != 0.75                                   # theta moved one RAdam step away from the file's init
    assert abs(float(mt.state[mt.n_lr]) - 0.75) < 1e-3
    out = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith('{')]
    assert {'parent_state': 'file', 'dataset_key': 'train'} in out
    # no path: the seeded synthetic state, and the log line says so
    train_meta.main(['with', 'YouTube-VOS', 'meta_batch_size=1', 'num_epochs.train=1', f'save_dir={tmp_path}', 'env_suffix=q'],
                    height=common.H, width=common.W, num_meta_iters=1, data_root=str(tmp_path / 'none'), eval_cmd=False, device='cpu')
    out = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith('{')]
    assert {'parent_state': 'synthetic', 'dataset_key': 'train'} in out
    with pytest.raises(NotImplementedError):                                  # train_meta.py:92-93
        train_meta.load_parent_state(None, {'train': {'states': [1, 2]}}, 'train', 'resnet50')


def test_eval_mode_loads_the_parent_state_of_each_dataset(tmp_path, fake_cli, capsys):
    train_meta, fake_engine, common = fake_cli
    sd = _parent_file(tmp_path / 'val_parent.model', -0.5)
    train_meta.main(['with', 'DAVIS-2017', 'e-OSVOS', 'num_epochs.eval=1', f'save_dir={tmp_path}', 'env_suffix=e',
                     'data_cfg.random_train_transform=False', f'parent_model.val.paths=[{tmp_path / "val_parent.model"}]'], height=common.H, width=common.W, num_frames=3,
                    data_root=str(tmp_path / 'none'), device='cpu')
    out = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith('{')]
    assert {'parent_state': 'file', 'dataset_key': 'val'} in out and {'parent_state': 'synthetic', 'dataset_key': 'train'} in out
    assert any(torch.equal(e.norm_args[2][:64], sd['backbone.bn1.running_mean']) for e in fake_engine.FakeEngine.instances
               if getattr(e, 'norm_args', None) is not None)


def test_resume_modes(tmp_path, fake_cli):
    train_meta, fake_engine, common = fake_cli
    from eosvos_amd.checkpoint import load_meta_checkpoint, save_meta_checkpoint
    assert train_meta.resume_checkpoint_name('LAST') == 'last_meta_iter.model'
    assert train_meta.resume_checkpoint_name('BEST_VAL') == 'best_val_meta_iter.model'        # train_meta.py:73-74
    with pytest.raises(NotImplementedError):
        train_meta.resume_checkpoint_name('NEWEST')
    args = ['with', 'YouTube-VOS', 'meta_batch_size=1', 'num_epochs.train=1', 'vis_interval=2', f'save_dir={tmp_path}', 'env_suffix=r']
    kw = dict(height=common.H, width=common.W, data_root=str(tmp_path / 'none'), eval_cmd=False, device='cpu')
    mt = train_meta.main(args, num_meta_iters=3, **kw)
    run = tmp_path / 'r'
    sd, info = load_meta_checkpoint(run / 'last_meta_iter.model')
    assert info['meta_iter'] == 3 and len(sd) == 128
    save_meta_checkpoint(str(run / 'best_val_meta_iter.model'), sd, 40, 1)
    mt2 = train_meta.main(args + ['resume_meta_run_epoch_mode=BEST_VAL'], num_meta_iters=2, **kw)
    _, info2 = load_meta_checkpoint(run / 'last_meta_iter.model')
    assert info2['meta_iter'] == 42                                         # continued from the BEST_VAL checkpoint
    with pytest.raises(FileNotFoundError):                                  # a missing resume file is an error, as in the reference
        train_meta.main(args[:-1] + ['env_suffix=fresh', 'resume_meta_run_epoch_mode=LAST'], num_meta_iters=1, **kw)


def test_checkpoint_cadence(tmp_path, fake_cli, monkeypatch):
    """`last_meta_iter.model` at meta-iteration 1, every `vis_interval` and at the end (train_meta.py:275-286)."""
    train_meta, fake_engine, common = fake_cli
    saved = []
    real = train_meta.save_meta_checkpoint
    monkeypatch.setattr(train_meta, 'save_meta_checkpoint', lambda path, sd, it, ep, *a: (saved.append(it), real(path, sd, it, ep, *a)))
    train_meta.main(['with', 'YouTube-VOS', 'meta_batch_size=1', 'num_epochs.train=1', 'vis_interval=3', f'save_dir={tmp_path}',
                     'env_suffix=c'], height=common.H, width=common.W, num_meta_iters=8, data_root=str(tmp_path / 'none'),
                    eval_cmd=False, device='cpu')
    assert saved == [1, 3, 6, 8]


def test_meta_training_runs_until_sigterm(tmp_path):
    """No iteration cap (`while True`, train_meta.py:207): SIGTERM finishes the iteration in flight, writes
    last_meta_iter.model and exits 0."""
    script = os.path.join(HERE, 'mp_workers', 'train_meta_forever.py')
    env = dict(os.environ, OMP_NUM_THREADS='2', WORLD_SIZE='1', RANK='0')
    p = subprocess.Popen([sys.executable, script, str(tmp_path)], env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    t0 = time.time()
    while time.time() - t0 < 300:
        l = p.stdout.readline()
        if l.startswith('{') and '"meta_iter"' in l:
            lines.append(json.loads(l))
            if len(lines) == 5:                       # well past any fixed small cap
                p.send_signal(signal.SIGTERM)
                break
    rest = p.communicate(timeout=300)[0]
    assert p.returncode == 0
    lines += [json.loads(l) for l in rest.splitlines() if l.startswith('{') and '"meta_iter"' in l]
    last = lines[-1]['meta_iter']
    assert last >= 5
    ck = torch.load(os.path.join(str(tmp_path), 'forever', 'last_meta_iter.model'), weights_only=False)
    assert ck['meta_iter'] == last


def test_validation_child_is_stopped_when_training_fails(tmp_path, fake_cli, monkeypatch):
    train_meta, fake_engine, common = fake_cli
    from eosvos_amd import meta_run
    procs = []
    real_popen = subprocess.Popen
    monkeypatch.setattr(train_meta.subprocess, 'Popen', lambda *a, **k: (procs.append(real_popen(*a, **k)), procs[-1])[1])

    def boom(self, *a, **k):
        raise RuntimeError('data error in the middle of meta-training')
    monkeypatch.setattr(meta_run.MetaTrainer, 'meta_iteration', boom)
    eval_cmd = [sys.executable, os.path.join(HERE, 'mp_workers', 'eval_child.py')]
    with pytest.raises(RuntimeError, match='data error'):
        train_meta.main(['with', 'YouTube-VOS', 'meta_batch_size=1', 'num_epochs.train=1', f'save_dir={tmp_path}', 'env_suffix=x'],
                        height=common.H, width=common.W, num_frames=3, num_meta_iters=2, data_root=str(tmp_path / 'none'),
                        eval_cmd=eval_cmd, device='cpu')
    assert len(procs) == 1 and procs[0].poll() is not None                 # the child was told to stop and has exited
    assert os.path.exists(tmp_path / 'x' / 'eval_stop')


def test_validation_child_exits_when_its_parent_is_gone(tmp_path):
    """The trainer was killed (no eval_stop): the child notices that its parent pid changed and leaves."""
    cfg = tmp_path / 'cfg.json'
    from eosvos_amd import config as config_mod
    json.dump(config_mod.parse_cli(['with', 'YouTube-VOS']), open(cfg, 'w'))
    child = os.path.join(HERE, 'mp_workers', 'eval_child.py')
    code = ("import os, subprocess, sys; p = subprocess.Popen([sys.executable, %r, '--run-dir', %r, '--config', %r, '--device', 'cpu', "
            "'--height', '16', '--width', '24', '--num-frames', '3', '--parent-pid', str(os.getpid())]); print(p.pid, flush=True)"
            % (child, str(tmp_path), str(cfg)))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, OMP_NUM_THREADS='2'))
    pid = int(out.stdout.split()[0])
    t0 = time.time()
    while time.time() - t0 < 120:
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        # a zombie re-parented to this test process's init still "exists": check its state
        try:
            if open(f'/proc/{pid}/stat').read().split()[2] == 'Z':
                break
        except FileNotFoundError:
            break
        time.sleep(0.5)
    else:
        os.kill(pid, signal.SIGKILL)
        pytest.fail('validation child kept running after its parent exited')


def test_late_objects_are_fine_tuned_on_their_first_annotated_frame(tmp_path, fake_cli):
    """YouTube-VOS object that first appears in frame 2: its train-frame mask comes from that frame, the frames before it
    stay background, the other object still trains on frame 0 (evaluate.py:132-168)."""
    train_meta, fake_engine, common = fake_cli
    from PIL import Image
    from eosvos_amd import config as config_mod, evaluate
    from eosvos_amd.data import open_dataset
    from eosvos_amd.meta_optim import MetaOptimizer
    root = tmp_path / 'YouTube-VOS'
    meta = {'videos': {}}
    rng = np.random.default_rng(2)
    objs = {'1': [0, 1, 2, 3, 4], '4': [2, 3, 4]}
    (root / 'train' / 'JPEGImages' / 'b01').mkdir(parents=True)
    (root / 'train' / 'Annotations' / 'b01').mkdir(parents=True)
    for f in range(5):
        Image.fromarray(rng.integers(0, 256, (16, 24, 3), dtype=np.uint8)).save(root / 'train' / 'JPEGImages' / 'b01' / f'{5 * f:05d}.jpg')
        lab = np.zeros((16, 24), np.uint8)
        lab[2:6, 2 + f:8 + f] = 1
        if f >= 2:
            lab[9:14, 10:20] = 4
        Image.fromarray(lab, mode='L').save(root / 'train' / 'Annotations' / 'b01' / f'{5 * f:05d}.png')
    meta['videos']['b01'] = {'objects': {k: {'category': 'x', 'frames': [f'{5 * f:05d}' for f in v]} for k, v in objs.items()}}
    (root / 'train' / 'meta.json').write_text(json.dumps(meta))
    (root / 'train_seqs.txt').write_text('b01\n')
    ds = open_dataset('YouTube-VOS', 'train_seqs', str(tmp_path), multi_object='single_id')
    frames, gts, fids = ds.sequence_tensors('b01', with_frame_ids=True)
    assert fids == [0, 2] and float(gts[0].sum()) == 24.0 and float(gts[1].sum()) == 50.0      # object 4's mask is NOT empty
    assert ds.frame_id is None and ds._label_id is None and ds.multi_object_id is None          # the reader is left as it was
    cfg = config_mod.parse_cli(['with', 'YouTube-VOS', 'num_epochs.eval=2'])
    cfg['datasets']['t'] = {'name': 'YouTube-VOS', 'split': 'train_seqs', 'eval': True}
    model, _ = train_meta.init_parent_model(**cfg['parent_model'])
    from eosvos_amd import synthetic
    model.load_state_dict(synthetic.synthetic_state('resnet50'))
    mo = MetaOptimizer(model, **cfg['meta_optim_cfg'])
    seen = []
    real = evaluate.finetune_object_steps

    def spy(model_, mo_, msd, fr, gt, cfg_, augment=None, train_frame_id=0):
        seen.append((train_frame_id, float(gt.sum())))
        return real(model_, mo_, msd, fr, gt, cfg_, augment, train_frame_id)
    evaluate.finetune_object_steps = spy
    try:
        res = evaluate.evaluate_dataset(model, mo, mo.state_dict(), ds, cfg, 't', save_dir=str(tmp_path / 'out'), device='cpu',
                                        objects_in_flight=1)
    finally:
        evaluate.finetune_object_steps = real
    assert seen == [(0, 24.0), (2, 50.0)]
    labels = res['labels']['b01']
    assert labels.shape == (5, 16, 24)
    assert not (labels[:2] == 2).any()                      # object 2 does not exist before its first annotated frame
    assert (labels[2][9:14, 10:20] == 2).all()              # its train frame is seeded with the annotation (2 * GT, :167-168)

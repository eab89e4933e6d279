"""G12: the online-adaptation schedule, pinned by RUNNING the reference's own `evaluate()`.

    python tests/golden/make_g12.py          (build container only: needs /root/reference)

`src/util/evaluate.py::evaluate` is one function (sequence loop, object loop, online-adaptation rounds, batch
composition, inference ranges, label merge, PNG and checkpoint output) that cannot run the DeepLab model at this commit
(SURVEY.md 3.5).  Here it runs UNMODIFIED on stand-ins for everything around the index logic: a 2-parameter model,
in-memory loaders that behave like `VOSDataset` + `EpochSampler` as far as `evaluate()` touches them, a `run_loader`
that returns prescribed probabilities, recorders for `imageio.imsave` / `torch.save` / `set_random_seeds`.  The event
log -- which frames and which pseudo-labels enter every adaptation batch, the seeds, the inference ranges, when the
first-step weights are restored, the merged label maps, file names -- is written to `g12_online_adapt.json`
(`evaluate.py:140-253,283-326,332-382`).  `tests/test_cpu_host.py` replays the product's loop against it.
"""
import json
import logging
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refshim  # noqa: E402

_refshim.install()

import util.evaluate as ev  # noqa: E402  (reference, unmodified)
from meta_optim.meta_optim import MetaOptimizer  # noqa: E402  (reference)

from g12_scenarios import SCENARIOS, frame_image, object_gt, prob_map  # noqa: E402

HW = (8, 12)


class _Done(Exception):
    pass


class TinyModel(nn.Module):
    """2 parameters; `forward` -> [logits] like the reference models (deeplabv3plus.py:301)."""

    def __init__(self, log):
        super().__init__()
        self.conv = nn.Conv2d(3, 1, 1)
        self.log = log

    def train_without_dropout(self):
        self.train()

    def forward(self, x):
        self.log.append(['forward', [round(float(v) * 100) for v in x[:, 0, 0, 0]]])
        return [self.conv(x)]

    def load_state_dict(self, sd, strict=True):
        self.log.append(['model.load_state_dict'])
        return super().load_state_dict(sd, strict)


class Dataset:
    """What `evaluate()` touches of `VOSDataset` (vos_dataset.py): sequence / object selection, the train frame,
    the propagated pseudo ground truth, `__getitem__` -> {'image', 'gt', 'file_name'}."""

    def __init__(self, sc, log, hw=None, frame_image=frame_image, object_gt=object_gt):
        self.sc, self.log = sc, log
        self.hw, self._frame_image, self._object_gt = hw or HW, frame_image, object_gt
        self.seqs_names = list(sc['seqs'])
        self.crop_size = 1
        self.transform = 'random_transform'
        self.multi_object_id = None
        self.propagate_frame_gt = None
        self.frame_id = None
        self.test_mode = False
        self.all_frames = False
        self.labels = []
        self.seq = None

    def set_seq(self, name):
        self.seq = name
        self.num_object_groups = self.sc['seqs'][name]['objects']
        self.num_objects = self.num_object_groups
        self.num_objects_in_group = 1
        self.set_gt_frame_id()

    def set_gt_frame_id(self):
        self.frame_id = self.sc['train_frame']

    def __len__(self):
        return self.sc['seqs'][self.seq]['frames']

    def __getitem__(self, i):
        obj = self.multi_object_id or 0
        gt = self._object_gt(self.seq, obj, self.hw)
        if self.propagate_frame_gt is not None:
            gt = torch.from_numpy(np.ascontiguousarray(self.propagate_frame_gt.transpose(2, 0, 1)))
        return {'image': self._frame_image(i, self.hw), 'gt': gt, 'file_name': f'{i:05d}'}


class TrainLoader:
    """DataLoader(batch_sampler=EpochSampler(...)): ONE batch per pass = the dataset's current frame x batch size
    (helper_func.py:250-336,521-545)."""

    def __init__(self, ds, bsz, log):
        self.dataset, self.bsz, self.log = ds, bsz, log

    def __iter__(self):
        d = self.dataset
        s = d[d.frame_id]
        tname = d.transform if isinstance(d.transform, str) else type(d.transform).__name__
        self.log.append(['train_batch', d.frame_id, d.propagate_frame_gt is not None, tname])
        yield {'image': torch.stack([s['image']] * self.bsz), 'gt': torch.stack([s['gt']] * self.bsz)}


class Sampler:
    indices = None


class Loader:
    def __init__(self, ds):
        self.dataset, self.sampler = ds, Sampler()


class TorchProxy:
    """`torch` as `evaluate()` sees it: cuda devices become the CPU, `save` is recorded."""

    def __init__(self, log, save_dir):
        self._log, self._save_dir = log, save_dir

    def __getattr__(self, n):
        return getattr(torch, n)

    def device(self, *_a, **_k):
        return torch.device('cpu')

    def save(self, obj, path):
        self._log.append(['torch.save', os.path.relpath(path, self._save_dir), sorted(obj.keys()), obj['meta_iter']])


class SharedDict(dict):
    def __getitem__(self, k):
        if k == 'meta_iter' and dict.__getitem__(self, k) is not None:
            raise _Done()
        return dict.__getitem__(self, k)


class Recorder:
    def __init__(self, log, save_dir, name):
        self.log, self.save_dir, self.name = log, save_dir, name

    def imsave(self, path, arr):
        a = np.asarray(arr)[..., 0]
        self.log.append(['imsave', os.path.relpath(path, self.save_dir), a.astype(int).tolist() if a.size < 4096 else a.astype(np.uint8)])


class Plt:
    """matplotlib stand-in: the debug figures (evaluate.py:384-425) are out of scope, only their paths are logged."""

    def __init__(self, log, save_dir):
        self.log, self.save_dir = log, save_dir

    def __getattr__(self, n):
        return lambda *a, **k: Plt(self.log, self.save_dir)

    def savefig(self, path, **_k):
        self.log.append(['debug_png', os.path.relpath(path, self.save_dir)])


def run(sc, hw=None, model_factory=None, meta_state=None, run_loader_factory=None, frame_image=frame_image, object_gt=object_gt,
        log_labels=True):
    """`evaluate()` on scenario `sc`.  Defaults = G12 (2-parameter model, prescribed probabilities).  make_g17.py passes the
    reference DeepLabV3Plus (`model_factory(log)`), its learned state (`meta_state`) and a `run_loader` that runs it
    (`run_loader_factory(log) -> run_loader`; the reference's own `run_loader` dereferences `model.rpn`)."""
    hw = hw or HW
    log = []
    save_dir = tempfile.mkdtemp()
    model_log = log
    cfg = {
        'seed': sc['seed'], 'loss_func': 'cross_entropy',
        'datasets': {'val': {'name': 'DAVIS-2017', 'split': 'val_seqs'}},
        'data_cfg': {'multi_object': 'all', 'batch_sizes': {'train': sc['batch'], 'test': 1, 'meta': 1}},
        'parent_model': {'architecture': 'DeepLabV3Plus'},
        'meta_optim_cfg': dict(init_lr=1e-3, learn_model_init=True, second_order_gradients=False,
                               lr_hierarchy_level='NEURON', use_log_init_lr=False, max_lr=None),
        'train_early_stopping_cfg': {'patience': None, 'min_loss_improv': 0.001},
        'eval_online_adapt': {'step': sc['step'], 'reset_model_mode': sc['reset_model_mode'],
                              'num_epochs': sc['ona_epochs'], 'min_prop': 0.5},
        'num_epochs': {'eval': sc['eval_epochs'], 'train': 5},
    }
    train_ds, test_ds, meta_ds = (Dataset(sc, log, hw, frame_image, object_gt) for _ in range(3))

    def data_loaders(_dataset, **_cfg):
        return TrainLoader(train_ds, sc['batch'], log), Loader(test_ds), Loader(meta_ds)

    def run_loader(model, loader, loss_func, img_save_dir=None, return_probs=False, start_targets=None):
        idx = list(loader.sampler.indices)
        obj = loader.dataset.multi_object_id
        log.append(['run_loader', idx, float(start_targets.sum())])
        probs = torch.stack([prob_map(loader.dataset.seq, obj, f, hw) for f in idx]) if idx else torch.zeros(0, 1, *hw)
        return None, None, probs, torch.zeros(len(idx), 1, 4)
    if run_loader_factory is not None:
        run_loader = run_loader_factory(log)

    orig = {k: getattr(ev, k) for k in ('init_parent_model', 'data_loaders', 'run_loader', 'eval_loader', 'eval_davis_seq',
                                        'imageio', 'plt', 'torch', 'set_random_seeds', 'compute_loss', 'MetaOptimizer')}
    real_compute_loss = orig['compute_loss']

    def compute_loss(name, out, gts, *a, **k):
        log.append(['loss', name, [float(g.sum()) for g in gts]])
        return real_compute_loss(name, out, gts, *a, **k)

    class LoggedMetaOptimizer(MetaOptimizer):              # the reference class; only the calls are logged
        def load_state_dict(self, sd, *a, **k):
            log.append(['mo.load_state_dict'])
            return super().load_state_dict(sd, *a, **k)

        def reset(self, keep_state=False):
            log.append(['mo.reset', bool(keep_state)])
            return super().reset(keep_state)

        def eval(self):
            log.append(['mo.eval'])
            return super().eval()

        def step(self, loss):
            log.append(['mo.step'])
            return super().step(loss)

    ev.init_parent_model = (lambda **kw: (model_factory(model_log), {})) if model_factory else (lambda **kw: (TinyModel(model_log), {}))
    ev.data_loaders = data_loaders
    ev.run_loader = run_loader
    ev.eval_loader = lambda *a, **k: (None, None, [0.0], None)
    ev.eval_davis_seq = lambda d, seq: {'J': {'mean': [0.5], 'recall': [0.5], 'decay': [0.0]},
                                        'F': {'mean': [0.5], 'recall': [0.5], 'decay': [0.0]}}
    ev.imageio = Recorder(log, save_dir, 'imageio')
    ev.plt = Plt(log, save_dir)
    ev.torch = TorchProxy(log, save_dir)
    ev.set_random_seeds = lambda s: (log.append(['seed', s]), orig['set_random_seeds'](s))[1]
    ev.compute_loss = compute_loss
    ev.MetaOptimizer = LoggedMetaOptimizer
    try:
        torch.manual_seed(0)
        msd = meta_state if meta_state is not None else MetaOptimizer(TinyModel([]), **cfg['meta_optim_cfg']).state_dict()
        shared = SharedDict(meta_iter=None, best_mean_J=0.0)
        try:
            ev.evaluate(0, 'val', msd, {'meta_iter': 3, 'meta_epoch': 1}, cfg, shared, save_dir, {}, True,
                        logging.getLogger('g12'))
        except _Done:
            pass
        result_keys = sorted(k for k in shared.keys())
    finally:
        for k, v in orig.items():
            setattr(ev, k, v)
    return {'scenario': sc, 'events': log, 'shared_dict_keys': result_keys,
            'time_per_frame_frames': sum(s['frames'] * s['objects'] for s in sc['seqs'].values())}


def main():
    out = {sc['name']: run(sc) for sc in SCENARIOS}
    with open(os.path.join(HERE, 'g12_online_adapt.json'), 'w') as f:
        json.dump(out, f)
    for name, r in out.items():
        kinds = {}
        for e in r['events']:
            kinds[e[0]] = kinds.get(e[0], 0) + 1
        print(name, kinds)


if __name__ == '__main__':
    main()

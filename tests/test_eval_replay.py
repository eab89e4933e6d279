"""A8 / G12: the product's evaluation worker replayed against the event log of the reference's own `evaluate()`
(`tests/golden/g12_online_adapt.json`, produced by `tests/golden/make_g12.py` running `src/util/evaluate.py`
UNMODIFIED on stand-in loaders): which frames and pseudo-labels enter every online-adaptation batch, the seeds,
the inference ranges, when the first-step weights are restored, the merged label maps, the prediction PNG paths and
the last/best checkpoint files.  CPU only: the engine is the test stand-in of tests/fake_engine.py.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
sys.path.insert(0, HERE)

from g12_scenarios import SCENARIOS, frame_image, object_gt, prob_map  # noqa: E402
from fake_engine import FakeDeepLab, FakeEngine  # noqa: E402

from eosvos_amd import config as config_mod  # noqa: E402
from eosvos_amd import evaluate as product_eval  # noqa: E402
from eosvos_amd.evaluate import online_adapt_schedule  # noqa: E402
from eosvos_amd.meta_optim import MetaOptimizer  # noqa: E402

HW = (8, 12)
G12 = json.load(open(os.path.join(HERE, 'golden', 'g12_online_adapt.json')))


def parse_reference(events, sc):
    """Reference event log -> {seq: {'objects': [[round, ...], ...], 'png': [(relpath, labels)]}}, 'saves'."""
    out, saves = {}, []
    rounds, cur, seed = [], {'theta': None, 'epochs': []}, None
    png = []
    for e in events:
        k = e[0]
        if k == 'mo.reset':
            cur['theta'] = 'init'                 # theta <- learned init
        elif k == 'model.load_state_dict':
            cur['theta'] = 'first_step'           # theta <- weights saved after round 0 (FIRST_STEP)
        elif k == 'seed':
            seed = e[1]
        elif k == 'forward':
            cur['epochs'].append([seed, e[1]])
        elif k == 'loss':
            cur['epochs'][-1].append([round(v, 3) for v in e[2]])
        elif k == 'run_loader':
            cur['infer'] = e[1]
            rounds.append(cur)
            cur = {'theta': None, 'epochs': []}
        elif k == 'imsave':
            png.append((e[1], e[2]))
        elif k == 'torch.save':
            saves.append((e[1], e[2], e[3]))
    # rounds -> sequences / objects by the frame counts of the scenario
    it = iter(rounds)
    pi = 0
    for seq, info in sc['seqs'].items():
        objs = []
        for _ in range(info['objects']):
            rs = []
            while True:
                r = next(it)
                rs.append(r)
                if not r['infer'] or r['infer'][-1] == info['frames'] - 1:
                    break
            objs.append(rs)
        out[seq] = {'objects': objs, 'png': png[pi:pi + info['frames']]}
        pi += info['frames']
    return out, saves


class ScenarioDataset:
    """The `eosvos_amd.data` reader interface `evaluate_dataset` uses, on the in-memory scenario."""
    test_mode = False

    def __init__(self, sc):
        self.sc = sc
        self.seqs_names = list(sc['seqs'])

    def sequence_tensors(self, seq, device='cpu'):
        info = self.sc['seqs'][seq]
        self.current = seq
        frames = torch.stack([frame_image(i, HW) for i in range(info['frames'])])
        return frames, [object_gt(seq, o, HW) for o in range(info['objects'])]

    def frame_names(self, seq):
        return [f'{i:05d}' for i in range(self.sc['seqs'][seq]['frames'])]

    def label_maps(self, seq):
        info = self.sc['seqs'][seq]
        lab = np.zeros((info['frames'],) + HW, np.uint8)
        for o in range(info['objects']):
            lab[:, object_gt(seq, o, HW)[0].numpy() > 0] = o + 1
        return lab


def run_product(sc, tmp_path, monkeypatch, objects_in_flight=None):
    cfg = config_mod.parse_cli([])
    cfg['seed'] = sc['seed']
    cfg['num_epochs']['eval'] = sc['eval_epochs']
    cfg['eval_online_adapt'].update(step=sc['step'] or 0, reset_model_mode=sc['reset_model_mode'],
                                    num_epochs=sc['ona_epochs'], min_prop=0.5)
    cfg['data_cfg']['batch_sizes']['train'] = sc['batch']
    cfg['datasets']['val'] = {'name': 'DAVIS-2017', 'split': 'val_seqs', 'eval': True}
    events = []
    model = FakeDeepLab('resnet50', num_classes=1, batch_norm=cfg['parent_model']['batch_norm'], max_batch=sc['batch'])
    model._views['backbone.conv1.weight'].view(-1)[0] = 0.3          # the stand-in network's two parameters
    meta_optim = MetaOptimizer(model, **cfg['meta_optim_cfg'])
    msd = meta_optim.state_dict()
    ds = ScenarioDataset(sc)

    real_loss, real_seed = product_eval.compute_loss, product_eval.set_random_seeds

    def compute_loss(name, out, gts, *a, **k):
        events.append(['loss', name, [round(float(g.sum()), 3) for g in gts]])
        return real_loss(name, out, gts, *a, **k)
    monkeypatch.setattr(product_eval, 'compute_loss', compute_loss)
    monkeypatch.setattr(product_eval, 'set_random_seeds', lambda s: (events.append(['seed', s]), real_seed(s))[1])

    real_call = FakeDeepLab.__call__

    def logged_call(self, inputs):
        out = real_call(self, inputs)           # may create the engine and push the state first (theta <- init)
        events.append(['forward', [round(float(v) * 100) for v in inputs[:, 0, 0, 0]]])
        self.engine.infer_fn = infer_fn
        return out
    monkeypatch.setattr(FakeDeepLab, '__call__', logged_call)
    for name, tag in (('reset', 'reset'), ('meta_task_begin', 'reset'), ('restore', 'restore'), ('set_init', 'reset')):
        real = getattr(FakeEngine, name)
        monkeypatch.setattr(FakeEngine, name, (lambda f, t: (lambda self, *a, **k: (events.append([t]), f(self, *a, **k))[1]))(real, tag))

    def infer_fn(eng, images):
        f = round(float(images[0, 0, 0, 0]) * 100)
        gt0 = eng.last_masks[0]
        obj = 0 if torch.equal(gt0, object_gt(ds.current, 0, HW)) else 1
        events.append(['infer', f])
        return prob_map(ds.current, obj, f, HW).view(1, 1, *HW)

    res = product_eval.evaluate_dataset(model, meta_optim, msd, ds, cfg, 'val', save_dir=str(tmp_path), meta_iter=3,
                                        meta_epoch=1, best_mean_J=0.0, objects_in_flight=objects_in_flight)
    if objects_in_flight:
        return events, res
    # events -> the same structure as parse_reference
    out = {}
    it = iter(events)
    evs = list(events)
    pos = 0
    for seq, info in sc['seqs'].items():
        n_rounds = len(online_adapt_schedule(info['frames'], sc['train_frame'], sc['step'] or 0, sc['batch']))
        objs = []
        for _ in range(info['objects']):
            rs = []
            for _r in range(n_rounds):
                cur = {'theta': None, 'epochs': [], 'infer': []}
                seed = None
                while pos < len(evs):
                    e = evs[pos]
                    if e[0] == 'infer':
                        cur['infer'].append(e[1])
                        pos += 1
                        if pos >= len(evs) or evs[pos][0] != 'infer':
                            break
                        continue
                    if e[0] == 'reset' and not cur['epochs']:
                        cur['theta'] = 'init'
                    elif e[0] == 'restore' and not cur['epochs']:
                        cur['theta'] = 'first_step'
                    elif e[0] == 'seed':
                        seed = e[1]
                    elif e[0] == 'forward':
                        cur['epochs'].append([seed, e[1]])
                    elif e[0] == 'loss':
                        cur['epochs'][-1].append(e[2])
                    pos += 1
                rs.append(cur)
            objs.append(rs)
        out[seq] = {'objects': objs}
    return out, res


@pytest.mark.parametrize('sc', SCENARIOS, ids=[s['name'] for s in SCENARIOS])
def test_evaluation_worker_replays_reference_event_log(sc, tmp_path, monkeypatch):
    ref, saves = parse_reference(G12[sc['name']]['events'], sc)
    got, res = run_product(sc, tmp_path, monkeypatch)
    for seq in sc['seqs']:
        for o, (r_obj, g_obj) in enumerate(zip(ref[seq]['objects'], got[seq]['objects'])):
            assert len(r_obj) == len(g_obj), (seq, o)
            for ri, (r, g) in enumerate(zip(r_obj, g_obj)):
                where = (sc['name'], seq, o, ri)
                assert g['infer'] == r['infer'], where                       # inference frame range of the round
                assert g['epochs'] == r['epochs'], where                     # seeds, batch frames, (pseudo) label sums
                # the weights every round starts from: the learned init, or (FIRST_STEP) those saved after round 0
                assert g['theta'] == r['theta'] == ('init' if ri == 0 or sc['reset_model_mode'] == 'FULL' else 'first_step'), where
        # merged label maps and prediction PNGs
        for f, (rel, lab) in enumerate(ref[seq]['png']):
            assert np.array_equal(res['labels'][seq][f].numpy(), np.array(lab, np.uint8)), (seq, f)
            path = os.path.join(str(tmp_path), rel)
            assert os.path.exists(path), rel
            assert np.array_equal(np.array(Image.open(path)), np.array(lab, np.uint8))
    # last / best checkpoints of the eval worker (evaluate.py:361-382)
    for rel, keys, meta_iter in saves:
        ck = torch.load(os.path.join(str(tmp_path), rel), map_location='cpu', weights_only=False)
        assert sorted(ck.keys()) == keys and ck['meta_iter'] == meta_iter
    assert {s[0] for s in saves} == {'last_val_meta_iter.model', 'best_val_meta_iter.model'}


def test_objects_in_flight_replay_the_same_work(tmp_path, monkeypatch):
    """`evaluate_dataset(objects_in_flight=2)`: the two objects of a sequence are fine-tuned side by side (one model /
    engine each, steps interleaved at the points where the host would wait for a loss).  Per object the events -- seeds,
    batches, pseudo-labels, restores, inference frames -- are those of the one-after-the-other run, and the merged
    label maps, J and checkpoints are identical."""
    sc = SCENARIOS[0]                                   # 'bear' has two objects, 'cows' one (falls back to sequential)
    with monkeypatch.context() as m:
        ev_seq, res_seq = run_product(sc, tmp_path / 'seq', m, objects_in_flight=1)
    with monkeypatch.context() as m:
        ev_con, res_con = run_product(sc, tmp_path / 'con', m, objects_in_flight=2)
    assert res_con['J_seq'] == res_seq['J_seq'] and res_con['mean_J'] == res_seq['mean_J']
    for seq in sc['seqs']:
        assert torch.equal(res_con['labels'][seq], res_seq['labels'][seq])
    # same multiset of events; the in-flight run interleaves the two objects of 'bear' (its forward batches alternate)
    # (theta <- init events excepted: every worker engine uploads the learned init once more when it is built, and the
    # in-flight run builds one engine per worker on top of the caller's; per-object resets are pinned by the replay above)
    key = lambda e: json.dumps(e)
    strip = lambda ev: sorted(map(key, (e for e in ev if e != ['reset'])))
    assert strip(ev_con) == strip(ev_seq)
    assert sum(e == ['reset'] for e in ev_con) >= sum(e == ['reset'] for e in ev_seq)
    assert [e for e in ev_con if e[0] != 'seed'] != [e for e in ev_seq if e[0] != 'seed']
    inf = [e[1] for e in ev_con if e[0] == 'infer'][:6]
    assert inf == [1, 2, 3, 1, 2, 3]                                     # one inference batch of object 0, one of object 1, ...
    for sub in ('seq', 'con'):
        assert os.path.exists(tmp_path / sub / 'best_val_meta_iter.model')


def test_more_objects_than_workers():
    """`run_objects_in_flight` with 2 workers and 5 objects: a worker that finishes takes the next object; every object's
    probabilities and loss history equal the one-after-the-other run."""
    from eosvos_amd.evaluate import finetune_object, object_workers, run_objects_in_flight
    cfg = config_mod.parse_cli([])
    cfg['num_epochs']['eval'] = 3
    cfg['eval_online_adapt'].update(step=3, reset_model_mode='FIRST_STEP', num_epochs=2, min_prop=0.5)
    cfg['data_cfg']['batch_sizes']['train'] = 3
    model = FakeDeepLab('resnet50', num_classes=1, batch_norm=cfg['parent_model']['batch_norm'], max_batch=3)
    model._views['backbone.conv1.weight'].view(-1)[0] = 0.3
    mo = MetaOptimizer(model, **cfg['meta_optim_cfg'])
    msd = mo.state_dict()
    frames = torch.stack([frame_image(i, HW) for i in range(8)])
    gts = [object_gt('s', o % 2, HW) * (1.0 if o < 2 else 0.0) + (torch.rand(1, *HW, generator=torch.Generator().manual_seed(o)) > 0.7).float() * (o >= 2)
           for o in range(5)]
    one = [finetune_object(model, mo, msd, frames, g, cfg) for g in gts]
    workers = object_workers(model, mo, cfg['meta_optim_cfg'], 2)
    assert len(workers) == 2 and all(w.model is not model for w in workers)
    con = run_objects_in_flight(workers, msd, frames, gts, cfg)
    assert len(con) == 5
    for (p2, h2), (p1, h1) in zip(con, one):
        assert h2 == h1 and torch.equal(p2, p1)
    assert len({round(float(h[0][-1]), 6) for _, h in one}) >= 3         # the objects really differ


def test_object_workers_follow_the_parent_checkpoint_of_each_dataset_key(tmp_path):
    """`evaluate()` loads the parent checkpoint of every dataset key into the model (`evaluate.py:46-50`).  The cached
    object workers are spawned copies: a second `evaluate_dataset` after `model.load_state_dict(other parent)` must run the
    multi-object sequences on the NEW weights and norm statistics (with `learn_model_init: False` the meta-optimizer
    state carries neither)."""
    sc = SCENARIOS[0]                                   # 'bear': two objects -> fine-tuned on the worker models
    cfg = config_mod.parse_cli([])
    cfg['num_epochs']['eval'] = 2
    cfg['eval_online_adapt'].update(step=0, reset_model_mode='FIRST_STEP', num_epochs=1, min_prop=0.5)
    cfg['data_cfg']['batch_sizes']['train'] = 2
    cfg['meta_optim_cfg']['learn_model_init'] = False
    cfg['datasets']['val'] = {'name': 'DAVIS-2017', 'split': 'val_seqs', 'eval': True}
    model = FakeDeepLab('resnet50', num_classes=1, batch_norm=cfg['parent_model']['batch_norm'], max_batch=2)
    model._views['backbone.conv1.weight'].view(-1)[0] = 0.3
    mo = MetaOptimizer(model, **cfg['meta_optim_cfg'])
    msd = mo.state_dict()
    assert not any(k.startswith('model_init_') for k in msd)
    ds = ScenarioDataset(sc)
    res_a = product_eval.evaluate_dataset(model, mo, msd, ds, cfg, 'val', objects_in_flight=2)
    workers = model._object_workers
    sd_b = model.state_dict()
    sd_b['backbone.conv1.weight'] = sd_b['backbone.conv1.weight'].clone()
    sd_b['backbone.conv1.weight'].view(-1)[0] = -0.7
    sd_b['backbone.bn1.running_var'] = sd_b['backbone.bn1.running_var'] * 3.0
    model.load_state_dict(sd_b)                          # the other dataset key's parent checkpoint
    res_b = product_eval.evaluate_dataset(model, mo, msd, ds, cfg, 'val', objects_in_flight=2)
    assert model._object_workers is workers              # the cache was reused ...
    for w in workers:                                    # ... and every worker engine now holds parent B
        assert float(w.model.engine.init[0]) == pytest.approx(-0.7)
        assert torch.allclose(w.model.engine.norm_args[3][:64], torch.full((64,), 3.0))
    one = FakeDeepLab('resnet50', num_classes=1, batch_norm=cfg['parent_model']['batch_norm'], max_batch=2)
    one.load_state_dict(sd_b)
    mo1 = MetaOptimizer(one, **cfg['meta_optim_cfg'])
    res_1 = product_eval.evaluate_dataset(one, mo1, msd, ds, cfg, 'val', objects_in_flight=1)
    for seq in sc['seqs']:
        assert torch.equal(res_b['labels'][seq], res_1['labels'][seq])
    assert res_b['J_seq'] == res_1['J_seq']


def test_schedule_function_matches_reference_rounds():
    """`online_adapt_schedule` alone against the inference ranges / propagated frames of the reference run."""
    for sc in SCENARIOS:
        ref, _ = parse_reference(G12[sc['name']]['events'], sc)
        for seq, info in sc['seqs'].items():
            sched = online_adapt_schedule(info['frames'], sc['train_frame'], sc['step'] or 0, sc['batch'])
            rounds = ref[seq]['objects'][0]
            assert [list(range(r['eval_min'], r['eval_max'])) for r in sched] == [r['infer'] for r in rounds]
            for r, rr in zip(sched[1:], rounds[1:]):
                # frames offered for propagation: the reference batch holds the train frame + the non-empty ones
                assert set(rr['epochs'][0][1][1:]) <= set(r['propagate_frames'])

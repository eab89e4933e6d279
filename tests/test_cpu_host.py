"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the
header declares (no compute calls), its topology table agrees with the Python mirror, the
golden layout and the oracle, the product never imports the oracle, and the N>1 outer-step
logic (task sharding + all-reduce + identical update on every rank) under gloo."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, 'e-osvos_amd', 'libeosvos.so')):
        ge.build()
    from eosvos_amd import _ffi
    return _ffi.load()


def test_header_symbols_exported(lib):
    from eosvos_amd import _ffi
    hdr = open(os.path.join(ROOT, 'include', 'eosvos.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = sorted(set(re.findall(r'\b(eosvos_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/eosvos.h but not exported'
    assert declared == _ffi.exported_symbols(), 'ctypes binding and header disagree'
    assert b'gfx950' in lib.eosvos_version()


def test_topology_c_vs_python_vs_golden(lib, golden_dir):
    from eosvos_amd import topology
    g = json.load(open(os.path.join(golden_dir, 'g1_layout.json')))
    for enc, arch in (('resnet50', 50), ('resnet101', 101)):
        infos = topology.conv_infos(enc)
        assert lib.eosvos_num_convs(arch) == len(infos) == len(g[enc + '_convs'])
        off = 0
        for i, c in enumerate(infos):
            buf = (ctypes.c_int64 * 9)()
            assert lib.eosvos_conv_info(arch, i, buf) == 0
            assert list(buf)[:8] == [c.cin, c.cout, c.k, c.stride, c.dil, c.pad, int(c.norm is not None), int(c.bias)]
            assert buf[8] == off
            off += c.cout * c.cin * c.k * c.k + (c.cout if c.bias else 0)
            assert g[enc + '_convs'][i] == [c.name, c.cin, c.cout, c.k, c.stride, c.dil, c.pad, c.bias]
        assert lib.eosvos_param_count(arch) == off
        assert lib.eosvos_lr_count(arch) == sum(s[0] for _, s in topology.trainable(enc))
        assert lib.eosvos_norm_count(arch) == sum(c for _, c in topology.norm_layers(enc))
    assert lib.eosvos_param_count(50) == 40289729 and lib.eosvos_lr_count(50) == 28658
    assert lib.eosvos_num_convs(7) == -1
    assert lib.eosvos_conv_info(50, 999, (ctypes.c_int64 * 9)()) != 0
    assert b'index' in lib.eosvos_last_error()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'e-osvos_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), fn
    for fn in os.listdir(os.path.join(pkg, 'csrc')):
        if not os.path.isfile(os.path.join(pkg, 'csrc', fn)):
            continue
        assert 'oracle' not in open(os.path.join(pkg, 'csrc', fn), errors='ignore').read().lower() or fn == 'Makefile'


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from eosvos_amd import _ffi
    from eosvos_amd.engine import Engine
    with pytest.raises(_ffi.EosvosError):
        Engine('resnet50', 96, 160, 1)


HIER_CASES = [('SINGLE', False), ('TENSOR', False), ('TENSOR', True), ('NEURON', True), ('PARAM', False),
              ('PARAM', True), ('SINGLE', True)]


@pytest.mark.parametrize('level,use_log', HIER_CASES)
def test_meta_optimizer_hierarchy_layout(golden_dir, level, use_log):
    """State-dict keys/shapes, init_lr, state_lr and clamp_init_lr of every lr_hierarchy_level /
    use_log_init_lr combination equal the reference MetaOptimizer's (fixture G13); host logic only."""
    from eosvos_amd import synthetic, topology
    from eosvos_amd.meta_optim import MetaOptimizer
    from eosvos_amd.networks import DeepLabV3Plus
    g = np.load(os.path.join(golden_dir, 'g13_lr_hierarchy.npz'))
    tag = f'{level}_{int(use_log)}'
    model = DeepLabV3Plus('resnet50', num_classes=1,
                          batch_norm={'accum_stats': False, 'learn_weight': False, 'learn_bias': False})
    mo = MetaOptimizer(model, init_lr=1e-3, learn_model_init=True, second_order_gradients=False,
                       lr_hierarchy_level=level, use_log_init_lr=use_log, max_lr=1.3e-3)
    assert [f'{k}:{tuple(v.shape)}' for k, v in mo.state_dict().items()] == list(g[tag + '_keys'])
    store = synthetic.synthetic_lr_state('resnet50', level, use_log)
    sd = synthetic.synthetic_state('resnet50')
    msd = {}
    if isinstance(store, list):
        for (n, _), lr in zip(topology.trainable('resnet50'), store):
            msd['log_init_lr_' + n.replace('.', '-')] = lr
    else:
        msd['log_init_lr'] = store
    for n, _ in topology.trainable('resnet50'):
        msd['model_init_' + n.replace('.', '-')] = sd[n]
    mo.load_state_dict(msd)
    np.testing.assert_allclose(mo.init_lr.numpy(), g[tag + '_init_lr'], rtol=1e-6)
    np.testing.assert_allclose(mo.state_lr.numpy(), g[tag + '_state_lr'], rtol=1e-6)
    mo.clamp_init_lr()
    t = mo._lr_flat.double()
    idx = torch.linspace(0, t.numel() - 1, 16).long()
    fp = np.concatenate([[t.sum().item(), t.norm().item()], t[idx].numpy()])
    np.testing.assert_allclose(fp, g[tag + '_clamped_fp'], rtol=1e-6)
    with pytest.raises(NotImplementedError):
        MetaOptimizer(model, init_lr=1e-3, learn_model_init=True, second_order_gradients=False,
                      lr_hierarchy_level='LAYER', use_log_init_lr=False, max_lr=None)


def test_meta_optimizer_without_learned_init(golden_dir):
    """`learn_model_init: False` registers only the lr tensors (meta_optim.py:76-78), fixture G13."""
    from eosvos_amd.meta_optim import MetaOptimizer
    from eosvos_amd.networks import DeepLabV3Plus
    g = np.load(os.path.join(golden_dir, 'g13_lr_hierarchy.npz'))
    model = DeepLabV3Plus('resnet50', num_classes=1,
                          batch_norm={'accum_stats': False, 'learn_weight': False, 'learn_bias': False})
    mo = MetaOptimizer(model, init_lr=1e-3, learn_model_init=False, second_order_gradients=False,
                       lr_hierarchy_level='NEURON', use_log_init_lr=False, max_lr=None)
    assert [f'{k}:{tuple(v.shape)}' for k, v in mo.state_dict().items()] == list(g['nolearn_keys'])
    assert [n for n, _ in mo.named_parameters()] == list(g['nolearn_named'])


def test_loop_helpers_vs_golden(golden_dir):
    """`early_stopping` (helper_func.py:388-397) and `EpochSampler` (:521-545) of the product against the outputs the
    reference's own functions produced (fixture G9/G10)."""
    from eosvos_amd.helper_func import EpochSampler, early_stopping
    g = json.load(open(os.path.join(golden_dir, 'g9_misc.json')))
    hist = [1.0, 0.5, 0.4, 0.3999, 0.3998, 0.3997, 0.39965]
    got = [[bool(early_stopping(hist[:n], p, 0.001)) for n in range(1, len(hist) + 1)] for p in (None, 2, 3)]
    assert got == g['early_stopping']
    assert [list(EpochSampler([0], False, 3)), list(EpochSampler([0, 1], False, 2))] == g['epoch_sampler']


def test_checkpoint_layout_compatible(golden_dir):
    """A `.model` file written by the reference's MetaOptimizer.state_dict() layout loads, and
    our state dict has the same key structure (train_meta.py:277-286)."""
    ck = torch.load(os.path.join(golden_dir, 'g11_last_meta_iter.model'), weights_only=False)
    assert set(ck) == {'meta_optim_state_dict', 'vis_win_names', 'meta_iter', 'meta_epoch'}
    keys = json.load(open(os.path.join(golden_dir, 'g11_keys.json')))['keys']
    assert [k for k, _ in keys] == list(ck['meta_optim_state_dict'])
    n = len(keys) // 2
    assert all(k.startswith('log_init_lr_') for k, _ in keys[:n])
    assert all(k.startswith('model_init_') for k, _ in keys[n:])
    from eosvos_amd.checkpoint import save_meta_checkpoint, load_meta_checkpoint
    path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'eosvos_ck_test.model')
    save_meta_checkpoint(path, ck['meta_optim_state_dict'], meta_iter=7, meta_epoch=2, vis_win_names={'dummy': 'w'})
    sd, info = load_meta_checkpoint(path)
    assert list(sd) == list(ck['meta_optim_state_dict']) and info['meta_iter'] == 7
    for k in sd:
        assert torch.equal(sd[k], ck['meta_optim_state_dict'][k])
    os.remove(path)


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
from eosvos_amd.meta_run import MetaTrainer, shard_tasks
from oracle import meta as ometa

class FakeEngine:
    """CPU stand-in with the Engine surface MetaTrainer uses; RAdam through the oracle."""
    encoder = 'resnet50'
    def __init__(self):
        self.n_lr, self.n_param, self.device = 8, 40, torch.device('cpu')
        self.states = {{}}
    def load_model_state(self, *a): pass
    def lr_store_count(self, level): return self.n_lr
    def set_loss(self, name): pass
    def set_lr_state(self, level, use_log, t): self.lr = t.clone()
    def set_init(self, t): self.init = t.clone()
    def meta_task_begin(self): pass
    def finetune_step(self, *a, **k): pass
    def meta_grad(self, xm, ym, flat, weight=1.0, init_grad=True, new_segment=False):
        flat += self.task_vec
        return float(self.task_loss)
    def radam_step(self, p, g, m, v, lr, wd, step, grad_scale=1.0, grad_clip=0.0):
        gg = g * grad_scale
        if grad_clip > 0: gg = gg.clamp(-grad_clip, grad_clip)
        st = dict(step=step - 1, exp_avg=m, exp_avg_sq=v)
        ometa.radam_step(p, gg, st, lr, wd)
    def clamp(self, p, lo, hi): p.clamp_(lo, hi)

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
eng = FakeEngine()
mt = MetaTrainer(eng, dist=dist, meta_batch_size=4, grad_clip=None)
g0 = torch.Generator().manual_seed(0)
mt.state.copy_(torch.rand(48, generator=g0))
tasks = [torch.randn(48, generator=torch.Generator().manual_seed(100 + t)) for t in range(4)]
mine = shard_tasks(4, rank, world)
for it in range(7):
    for t in mine:
        eng.task_vec, eng.task_loss = tasks[t] * (it + 1), (float('nan') if (t == 3 and it == 2) else 0.5)
        mt.run_task(None, None, None, None, inner_steps=0)
    dist.all_reduce(mt.grad)
    mt.outer_step()
torch.save(dict(state=mt.state, mine=mine, skipped=mt.skipped_tasks), {out!r} + f'.{{rank}}')
dist.destroy_process_group()
'''


def test_outer_step_two_ranks_gloo(tmp_path):
    """world_size 2 on CPU: sharded tasks + all-reduce(sum) + the same RAdam step on every
    rank == the single-process result; NaN-skipped task contributes zeros."""
    out = str(tmp_path / 'res')
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', OMP_NUM_THREADS='1')
    for world in (2, 1):
        procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE=str(world)))
                 for r in range(world)]
        for p in procs:
            assert p.wait(timeout=300) == 0
        res = [torch.load(out + f'.{r}', weights_only=False) for r in range(world)]
        if world == 2:
            assert res[0]['mine'] == [0, 2] and res[1]['mine'] == [1, 3]
            assert torch.equal(res[0]['state'], res[1]['state'])        # identical update, no broadcast
            assert res[0]['skipped'] + res[1]['skipped'] == 1
            two = res[0]['state']
        else:
            np.testing.assert_allclose(two.numpy(), res[0]['state'].numpy(), rtol=1e-6, atol=1e-8)
            assert float(res[0]['state'][:8].min()) >= 0


def test_config_accepts_every_reference_key(golden_dir):
    """G16: every key of cfgs/meta.yaml + cfgs/torch.yaml exists in config.BASE (Sacred rejects unknown keys, so a
    `with key=val` of the reference must stay valid) and the four named configs carry exactly the reference's values."""
    import json
    from eosvos_amd import config
    g = json.load(open(os.path.join(golden_dir, 'g16_cfg_keys.json')))

    def flat(d, p=''):
        out = {}
        for k, v in d.items():
            if isinstance(v, dict) and v:
                out.update(flat(v, p + k + '.'))
            else:
                out[p + k] = v
        return out
    base = flat(config.BASE)
    assert sorted(base) == sorted(g['base_keys'])
    for key in g['base_keys']:
        if key.startswith('data_cfg.crop_sizes.'):              # a known key whose non-null values are refused (not implemented)
            with pytest.raises(NotImplementedError):
                config.parse_cli(['with', f'{key}=1'])
            continue
        cfg = config.parse_cli(['with', f'{key}=1'])           # the CLI grammar accepts an override of any of them
        assert flat(cfg)[key] == 1
    for name, vals in g['named'].items():
        assert flat(config.NAMED[name]) == vals, name
        cfg = flat(config.parse_cli(['with', name]))
        for k, v in vals.items():
            assert cfg[k] == v, (name, k)
    with pytest.raises(KeyError):
        config.parse_cli(['with', 'not_a_reference_key=1'])


def test_collective_matrix_mode_verdict_reaches_every_engine(monkeypatch):
    """`MetaTrainer._collective_mode_check` (round 5): the f16x3 range guard runs once, on the first engine with the rank's first task,
    at the first meta-iteration after `load_state`; its verdict moves EVERY engine of the trainer (and, all-reduced, of every rank)
    -- engines of one trainer in different modes would average gradients of slightly different functions.  Host logic only: a
    stand-in engine with the product engine's guard attributes."""
    from eosvos_amd import synthetic
    from eosvos_amd.meta_run import MetaTrainer
    from fake_engine import FakeEngine

    class GuardedFake(FakeEngine):
        verdict = 'f16x3'

        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self._mode, self._verify_pending, self._step_check_pending, self.verified_with = 'f16x3', True, True, None

        @property
        def matrix_mode(self):
            return self._mode

        def set_engine_matrix_mode(self, mode):
            self._mode = mode

        def verify_matrix_mode(self, images, masks=None, loss_kind=None):
            self.verified_with = (images, masks)
            self._verify_pending = False
            if GuardedFake.verdict != 'f16x3':
                self._mode = GuardedFake.verdict
            return self._mode

        def load_model_state(self, sd, lrs=None):
            super().load_model_state(sd, lrs)
            self._verify_pending = True

    sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
    x, y = torch.zeros(1, 3, 8, 8), torch.zeros(1, 1, 8, 8)
    for verdict in ('f16x3', 'bf16x6'):
        GuardedFake.verdict = verdict
        engines = [GuardedFake('resnet50', 8, 8, 1) for _ in range(3)]
        mt = MetaTrainer(engines[0], meta_batch_size=3, extra_engines=engines[1:])
        mt.load_state(sd, lrs)
        assert mt._mode_check_pending
        flag = mt._collective_mode_check([(x, y, x, y)])
        assert flag == (verdict != 'f16x3') and not mt._mode_check_pending
        assert engines[0].verified_with[0] is x and engines[0].verified_with[1] is y      # the step half needs the masks
        assert engines[1].verified_with is None                                           # once per trainer, not per engine
        assert [e.matrix_mode for e in engines] == [verdict] * 3
        assert not any(e._verify_pending or e._step_check_pending for e in engines)
    # a rank without a task this iteration still takes part (flag 0) and keeps its engines' own lazy check for its first task
    GuardedFake.verdict = 'bf16x6'
    eng = GuardedFake('resnet50', 8, 8, 1)
    mt = MetaTrainer(eng, meta_batch_size=1)
    mt.load_state(sd, lrs)
    assert mt._collective_mode_check([]) == 0 and eng.matrix_mode == 'f16x3' and eng._verify_pending


def test_unimplemented_options_raise_instead_of_being_ignored():
    """Keys of cfgs/meta.yaml are all accepted (fixture G16); the ones this build does not implement must raise."""
    from eosvos_amd import config
    config.parse_cli(['with', 'DAVIS-2017', 'e-OSVOS'])
    with pytest.raises(NotImplementedError):
        config.parse_cli(['with', 'DAVIS-2017', 'data_cfg.crop_sizes.train=[256, 256]'])

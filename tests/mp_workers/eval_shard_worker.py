"""`evaluate_dataset` with the (sequence, object) work items dealt over the ranks (gloo): usage eval_shard_worker.py OUT SAVE_DIR"""
import os
import sys

import torch
import torch.distributed as dist

import common
from g12_scenarios import SCENARIOS, object_gt, prob_map
from test_eval_replay import HW, ScenarioDataset

from eosvos_amd import config as config_mod
from eosvos_amd import evaluate as product_eval
from eosvos_amd.meta_optim import MetaOptimizer

out, save_dir = sys.argv[1], sys.argv[2]
world = int(os.environ.get('WORLD_SIZE', 1))
if world > 1:
    dist.init_process_group('gloo')
sc = SCENARIOS[0]
cfg = config_mod.parse_cli([])
cfg['seed'] = sc['seed']
cfg['num_epochs']['eval'] = sc['eval_epochs']
cfg['eval_online_adapt'].update(step=sc['step'], reset_model_mode=sc['reset_model_mode'], num_epochs=sc['ona_epochs'])
cfg['data_cfg']['batch_sizes']['train'] = sc['batch']
cfg['datasets']['val'] = {'name': 'DAVIS-2017', 'split': 'val_seqs', 'eval': True}
model = common.FakeDeepLab('resnet50', num_classes=1, batch_norm=cfg['parent_model']['batch_norm'], max_batch=sc['batch'])
model._views['backbone.conv1.weight'].view(-1)[0] = 0.3
mo = MetaOptimizer(model, **cfg['meta_optim_cfg'])
ds = ScenarioDataset(sc)
items = []


def infer_fn(eng, images):
    f = round(float(images[0, 0, 0, 0]) * 100)
    obj = 0 if torch.equal(eng.last_masks[0], object_gt(ds.current, 0, HW)) else 1
    items.append((ds.current, obj))
    return prob_map(ds.current, obj, f, HW).view(1, 1, *HW)


real_call = common.FakeDeepLab.__call__


def call(self, x):
    o = real_call(self, x)
    self.engine.infer_fn = infer_fn
    return o


common.FakeDeepLab.__call__ = call
res = product_eval.evaluate_dataset(model, mo, mo.state_dict(), ds, cfg, 'val', save_dir=save_dir, meta_iter=1, meta_epoch=0,
                                    dist=dist if world > 1 else None, device='cpu')
torch.save({'labels': res['labels'], 'J_seq': res['J_seq'], 'items': sorted(set(items))}, f'{out}.{os.environ.get("RANK", "0")}')
if world > 1:
    dist.destroy_process_group()

"""G16: the key set of the reference's Sacred configuration (build container only).

    python tests/golden/make_cfg_keys.py
Flattens cfgs/meta.yaml + cfgs/torch.yaml (the base config, train_meta.py:22-23) into dotted keys with the YAML
type of each value, and the four named configs (`:24-27`) with their values, into g16_cfg_keys.json.
tests/test_cpu_host.py checks that eosvos_amd.config accepts exactly this key set.
"""
import json
import os

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
CFGS = '/root/reference/cfgs'


def flatten(d, prefix=''):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flatten(v, prefix + k + '.'))
        else:
            out[prefix + k] = v
    return out


base = {}
for f in ('meta.yaml', 'torch.yaml'):
    base.update(flatten(yaml.safe_load(open(os.path.join(CFGS, f)))))
named = {n: flatten(yaml.safe_load(open(os.path.join(CFGS, f)))) for n, f in (
    ('DAVIS-2017', 'meta_davis-2017.yaml'), ('YouTube-VOS', 'meta_youtube-vos.yaml'), ('e-OSVOS', 'eval_e-osvos.yaml'),
    ('e-OSVOS-OnA', 'eval_e-osvos-OnA.yaml'))}
json.dump({'base_keys': {k: type(v).__name__ for k, v in sorted(base.items())}, 'named': named},
          open(os.path.join(HERE, 'g16_cfg_keys.json'), 'w'), indent=1)
print(len(base), 'base keys;', {n: len(v) for n, v in named.items()})

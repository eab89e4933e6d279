"""evaluate_dataset on a DAVIS-2017 tree of 480x854 JPEGs in /tmp (4 sequences x 40 frames, 2 objects, 20 fine-tune
iterations per object, PNG output): seconds per sequence with the next sequence decoded / the last one written on worker
threads, and with everything on the calling thread."""
import json, os, subprocess, sys
import numpy as np
from PIL import Image
root = '/tmp/eosvos_evalfeed/data/DAVIS-2017'
if not os.path.isdir(root):
    rng = np.random.default_rng(0)
    for s in range(4):
        seq = f'seq{s}'
        os.makedirs(f'{root}/JPEGImages/480p/{seq}'); os.makedirs(f'{root}/Annotations/480p/{seq}')
        base = np.kron(rng.integers(0, 256, (30, 54, 3), dtype=np.uint8), np.ones((16, 16, 1), np.uint8))[:480, :854]
        for f in range(40):
            Image.fromarray(np.roll(base, 6 * f, axis=1)).save(f'{root}/JPEGImages/480p/{seq}/{f:05d}.jpg', quality=90)
            lab = np.zeros((480, 854), np.uint8); lab[100:220, 200 + 6 * f:400 + 6 * f] = 1; lab[300:420, 500 - 4 * f:700 - 4 * f] = 2
            Image.fromarray(lab, mode='L').save(f'{root}/Annotations/480p/{seq}/{f:05d}.png')
    open(f'{root}/val_seqs.txt', 'w').write(''.join(f'seq{s}\n' for s in range(4)))
code = '''
import sys, time, json, torch
sys.path.insert(0, ".")
from eosvos_amd import config, data, synthetic
from eosvos_amd import evaluate as ev
from eosvos_amd.helper_func import init_parent_model
from eosvos_amd.meta_optim import MetaOptimizer
cfg = config.parse_cli(["with", "DAVIS-2017", "e-OSVOS", "num_epochs.eval=20"])
cfg["datasets"]["val"] = dict(cfg["datasets"].get("val", {}), name="DAVIS-2017", split="val_seqs", eval=True)
ds = data.open_dataset("DAVIS-2017", "val_seqs", "/tmp/eosvos_evalfeed/data", multi_object="single_id")
model, _ = init_parent_model(**dict(cfg["parent_model"])); model.to("cuda:0")
model.load_state_dict(synthetic.synthetic_state(cfg["parent_model"]["encoder"]))
mo = MetaOptimizer(model, **cfg["meta_optim_cfg"]); msd = mo.state_dict()
warm = data.SyntheticSequences(1, 3, 480, 854)
ev.evaluate_dataset(model, mo, msd, warm, dict(cfg, num_epochs=dict(cfg["num_epochs"], eval=2)), "val")
torch.cuda.synchronize(); t0 = time.time()
res = ev.evaluate_dataset(model, mo, msd, ds, cfg, "val", save_dir="/tmp/eosvos_evalfeed/run")
torch.cuda.synchronize()
print(json.dumps({"seconds_per_sequence": round((time.time() - t0) / 4, 3), "ms_per_object_frame": round(1e3 * res["time_per_frame"], 2), "mean_J": round(res["mean_J"], 4)}))
'''
for pre in ('1', '0'):
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, EOSVOS_EVAL_PREFETCH=pre), capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    print(json.dumps(dict(json.loads(line[-1]), prefetch=pre == '1')) if line else out.stderr[-800:], flush=True)

"""Meta-training on a DAVIS-2017 tree of 480x854 JPEGs written to /tmp: seconds per meta-iteration (4 tasks per GPU)
with the decode + colour jitter of the next sub-batch prefetched on a worker thread, and without."""
import json, os, subprocess, sys, time
import numpy as np
from PIL import Image
root = '/tmp/eosvos_feed/data/DAVIS-2017'
if not os.path.isdir(root):
    rng = np.random.default_rng(0)
    for s in range(6):
        seq = f'seq{s}'
        os.makedirs(f'{root}/JPEGImages/480p/{seq}'); os.makedirs(f'{root}/Annotations/480p/{seq}')
        base = np.kron(rng.integers(0, 256, (30, 54, 3), dtype=np.uint8), np.ones((16, 16, 1), np.uint8))[:480, :854]
        for f in range(12):
            Image.fromarray(np.roll(base, 6 * f, axis=1)).save(f'{root}/JPEGImages/480p/{seq}/{f:05d}.jpg', quality=90)
            lab = np.zeros((480, 854), np.uint8); lab[150:330, 200 + 6 * f:500 + 6 * f] = 1
            Image.fromarray(lab, mode='L').save(f'{root}/Annotations/480p/{seq}/{f:05d}.png')
    open(f'{root}/train_seqs.txt', 'w').write(''.join(f'seq{s}\n' for s in range(6)))
code = '''
import sys, time, json
sys.path.insert(0, ".")
from eosvos_amd import train_meta
import io, contextlib
buf = io.StringIO(); t0 = time.time()
with contextlib.redirect_stdout(buf):
    train_meta.main(["with", "DAVIS-2017", "meta_batch_size=4", "datasets.train.eval=False", "save_dir=/tmp/eosvos_feed/run", "env_suffix=f"],
                    num_meta_iters=14, data_root="/tmp/eosvos_feed/data", eval_cmd=False)
lines = [l for l in buf.getvalue().splitlines() if l.startswith("{")]
print(json.dumps({"iterations": len(lines), "seconds_total": round(time.time() - t0, 2)}))
'''
for pre in ('1', '0'):
    env = dict(os.environ, EOSVOS_META_PREFETCH=pre)
    ts = []
    out = subprocess.run([sys.executable, '-c', code.replace('num_meta_iters=14', 'num_meta_iters=4')], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    a = json.loads(out)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    b = json.loads(out)
    per = (b['seconds_total'] - a['seconds_total']) / 10
    print(json.dumps({'prefetch': pre == '1', 'seconds_per_meta_iteration': round(per, 4), 'tasks_per_second': round(4 / per, 2)}), flush=True)

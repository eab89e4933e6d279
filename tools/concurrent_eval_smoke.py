"""BASELINE configs[4] on one GPU at a small frame size: meta-train iterations with the validation CHILD PROCESS running
beside them on the same device (it is spawned before this process touches the GPU).
    python tools/concurrent_eval_smoke.py [save_dir]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eosvos_amd import train_meta  # noqa: E402

save_dir = sys.argv[1] if len(sys.argv) > 1 else '/tmp/eosvos_smoke'
mt = train_meta.main(['with', 'YouTube-VOS', 'meta_batch_size=2', 'num_epochs.train=2', 'num_epochs.eval=3', 'vis_interval=1',
                      f'save_dir={save_dir}', 'env_suffix=smoke'], height=96, width=160, num_frames=4, num_meta_iters=3,
                     data_root=os.path.join(save_dir, 'no_data'))
run = os.path.join(save_dir, 'smoke')
lines = [json.loads(l) for l in open(os.path.join(run, 'eval_log.jsonl'))]
print('eval log:', lines)
assert lines and lines[-1]['meta_iter'] == 3
assert os.path.exists(os.path.join(run, 'last_val_davis17_meta_iter.model')) and os.path.exists(os.path.join(run, 'last_meta_iter.model'))
print('concurrent eval smoke ok: %d snapshot(s) evaluated beside %d meta-iterations' % (len(lines), mt.step))

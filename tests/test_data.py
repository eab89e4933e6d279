"""Host feeder: DAVIS-layout reader against the tensor contract of row A0, on a synthetic tree."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from eosvos_amd.data import DAVIS, jaccard, sequence_J


@pytest.fixture()
def davis_root(tmp_path):
    root = tmp_path / 'DAVIS-2017'
    rng = np.random.default_rng(0)
    for seq, n_obj in (('bear', 1), ('dogs', 2)):
        (root / 'JPEGImages' / '480p' / seq).mkdir(parents=True)
        (root / 'Annotations' / '480p' / seq).mkdir(parents=True)
        for f in range(4):
            img = rng.integers(0, 256, (48, 80, 3), dtype=np.uint8)
            Image.fromarray(img).save(root / 'JPEGImages' / '480p' / seq / f'{f:05d}.jpg', quality=95)
            lab = np.zeros((48, 80), np.uint8)
            lab[10:20, 10 + f:30 + f] = 1
            if n_obj == 2:
                lab[30:40, 40:60] = 2
            p = Image.fromarray(lab, mode='P')
            p.putpalette([0, 0, 0, 128, 0, 0, 0, 128, 0] + [0] * 759)
            p.save(root / 'Annotations' / '480p' / seq / f'{f:05d}.png')
        (root / 'JPEGImages' / '480p' / seq / '.hidden').write_text('x')
    (root / 'val_seqs.txt').write_text('bear\ndogs\n')
    return str(root)


def test_reader_contract(davis_root):
    ds = DAVIS('val_seqs', davis_root, multi_object='single_id')
    assert ds.seqs_names == ['bear', 'dogs'] and len(ds) == 8 and ds.year == 2017
    ds.set_seq('dogs')
    assert ds.num_objects == 2 and len(ds) == 4
    ds.multi_object_id = 1
    s = ds[0]
    assert s['image'].shape == (3, 48, 80) and s['image'].dtype == torch.float32
    assert 0.0 <= float(s['image'].min()) and float(s['image'].max()) <= 1.0            # /255, no mean subtraction
    assert s['gt'].shape == (1, 48, 80) and set(s['gt'].unique().tolist()) == {0.0, 1.0}
    assert float(s['gt'][0, 30:40, 40:60].min()) == 1.0 and float(s['gt'][0, 10:20].max()) == 0.0   # object 2 only
    ds.frame_id = 2
    assert len(ds) == 1 and ds[0]['file_name'] == '00002'
    single = DAVIS('bear', davis_root)                                                     # single-object mode
    assert single.num_objects == 1 and set(np.unique(single.make_img_label_pair(0)[1])) == {0.0, 1.0}
    frames, gts = ds.sequence_tensors('dogs')
    assert frames.shape == (4, 3, 48, 80) and len(gts) == 2 and gts[0].shape == (1, 48, 80)
    assert float(gts[0].sum()) == 200.0 and float(gts[1].sum()) == 200.0
    with pytest.raises(NotImplementedError):
        DAVIS('bear', davis_root, multi_object='per_pixel')


def test_jaccard():
    a = np.zeros((4, 4), bool); b = np.zeros((4, 4), bool)
    assert jaccard(a, b) == 1.0
    a[:2] = True; b[1:3] = True
    assert jaccard(a, b) == pytest.approx(4 / 12)
    labels = np.zeros((4, 4, 4), np.uint8); labels[:, :2] = 1
    assert sequence_J(labels, labels, 1) == 1.0


@pytest.fixture()
def youtube_root(tmp_path):
    """YouTube-VOS layout: root/train/{JPEGImages,Annotations}/<seq>/, root/train/meta.json, root/train_seqs.txt; object ids
    are not 1..n and the second object of 'b01' first appears in frame 2."""
    import json
    root = tmp_path / 'YouTube-VOS'
    rng = np.random.default_rng(1)
    meta = {'videos': {}}
    for seq, objs in (('a00', {'3': [0, 1, 2, 3, 4]}), ('b01', {'1': [0, 1, 2, 3, 4], '4': [2, 3, 4]})):
        (root / 'train' / 'JPEGImages' / seq).mkdir(parents=True)
        (root / 'train' / 'Annotations' / seq).mkdir(parents=True)
        for f in range(5):
            Image.fromarray(rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)).save(root / 'train' / 'JPEGImages' / seq / f'{5 * f:05d}.jpg')
            lab = np.zeros((40, 64), np.uint8)
            for k, frames in objs.items():
                if f in frames:
                    o = int(k)
                    lab[4 * o:4 * o + 8, 6 * o + f:6 * o + 12 + f] = o
            Image.fromarray(lab, mode='L').save(root / 'train' / 'Annotations' / seq / f'{5 * f:05d}.png')
        meta['videos'][seq] = {'objects': {k: {'category': 'x', 'frames': [f'{5 * f:05d}' for f in v]} for k, v in objs.items()}}
    (root / 'train' / 'meta.json').write_text(json.dumps(meta))
    (root / 'train_seqs.txt').write_text('a00\nb01\n')
    return str(tmp_path)


def test_youtube_reader_and_meta_taskset(youtube_root):
    import random
    from eosvos_amd import config as config_mod
    from eosvos_amd.data import YouTube, open_dataset
    from eosvos_amd.meta_tasksets import ColorJitterParams, ConcatTaskset, MetaTaskset, task_order
    ds = open_dataset('YouTube-VOS', 'train_seqs', youtube_root, multi_object='single_id')
    assert isinstance(ds, YouTube) and ds.seqs_names == ['a00', 'b01'] and not ds.test_mode and not ds.all_frames
    ds.set_seq('b01')
    assert ds.num_objects == 2 and ds._multi_object_id_to_label == [1, 4] and ds.num_object_groups == 2
    assert ds.get_gt_frame_id(0) == (0, 0) and ds.get_gt_frame_id(1) == (2, 2)      # object 4 first annotated in frame 2
    ds.multi_object_id = 1
    ds.set_gt_frame_id()
    assert (ds.frame_id, ds._label_id) == (2, 2)
    ds._label_id = None
    assert ds.make_img_label_pair(0)[1].sum() == 0 and ds.make_img_label_pair(3)[1].sum() == 8 * 12   # id 4 -> binary mask
    assert not ds.has_frame_object(0) and ds.has_frame_object(3)
    cfg = config_mod.parse_cli(['with', 'YouTube-VOS'])
    ts = MetaTaskset(ds, cfg['data_cfg'], random_frame_transform_per_task=True, single_obj_seq_mode='KEEP')
    assert ts.object_groups == [('a00', 0), ('b01', 0), ('b01', 1)]
    assert MetaTaskset(ds, cfg['data_cfg'], single_obj_seq_mode='IGNORE').object_groups == [('b01', 0), ('b01', 1)]
    torch.manual_seed(3)
    random.seed(3)
    for _ in range(6):
        item = ts[2]                                        # object 4 of b01: only frames 2..4 carry it
        assert item['seq_name'] == 'b01' and item['obj_id'] == 1 and item['train_frame'] in (2, 3, 4)
        assert all(f in (2, 3, 4) for f in item['meta_frames']) and len(item['meta_frames']) == 1
        tr = item['transform']
        assert isinstance(tr['flip'], bool) and sorted(n for n, _ in tr['color'].ops) == ['brightness', 'contrast', 'hue', 'saturation']
    ts_eps = MetaTaskset(ds, cfg['data_cfg'], random_frame_transform_per_task=False, random_frame_epsilon=5)
    for _ in range(6):
        item = ts_eps[0]
        assert abs(item['meta_frames'][0] - item['train_frame']) <= 1 and item['transform'] is None    # 5-frame stride
    with pytest.raises(NotImplementedError):
        MetaTaskset(ds, cfg['data_cfg'], random_box_coord_perm=True)
    # colour jitter: factors within the torchvision ranges, image stays in [0, 1]
    cj = ColorJitterParams(.2, .2, .2, .1, rng=random.Random(0))
    img = np.random.default_rng(0).random((8, 8, 3)).astype(np.float32)
    out = cj(img)
    assert out.shape == img.shape and 0.0 <= out.min() and out.max() <= 1.0 and np.abs(out - img).max() < 0.5
    for name, f in cj.ops:
        assert (-0.1 <= f <= 0.1) if name == 'hue' else (0.8 <= f <= 1.2)
    # the host half of a task (decode + colour jitter), as train_meta prefetches it on a worker thread: same pixels as the
    # direct path, the shared dataset object is left alone, and without a per-task transform the whole task materialises
    # from it (no engine needed)
    from concurrent.futures import ThreadPoolExecutor
    item = ts[2]
    ds.set_seq('a00')
    ds.multi_object_id = 0
    with ThreadPoolExecutor(1) as pool:
        host = pool.submit(ts.load_frames, item).result()
    assert ds.seq_key == 'a00' and ds.multi_object_id == 0
    assert sorted(host) == sorted({item['train_frame'], *item['meta_frames']})
    ds.set_seq('b01')
    ds.multi_object_id = 1
    for f, (img_h, lab_h) in host.items():
        img_d, lab_d = ds.make_img_label_pair(f)
        assert np.array_equal(img_h, item['transform']['color'](img_d)) and np.array_equal(lab_h, lab_d)
    it0 = ts_eps[1]
    got = ts_eps.task_tensors(it0, None, 'cpu', host=ts_eps.load_frames(it0))
    ref = ts_eps.task_tensors(it0, None, 'cpu')
    assert all(torch.equal(a, b) for a, b in zip(got, ref)) and got[0].shape[0] == cfg['data_cfg']['batch_sizes']['train']
    # a worker's shuffled sub-batches cover every task once per pass
    both = ConcatTaskset([ts, ts_eps])
    assert len(both) == 6 and both.locate(4) == (ts_eps, 1)
    order = task_order(len(both), 4, seed=5, epoch=0)
    assert sorted(i for b in order for i in b) == list(range(6)) and [len(b) for b in order] == [4, 2]
    assert order != task_order(len(both), 4, seed=6, epoch=0)

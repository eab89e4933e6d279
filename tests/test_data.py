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

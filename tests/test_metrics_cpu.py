"""rscotr_amd.metrics / the datasets' evaluate(): top-k accuracy, confusion-matrix segmentation metrics and
COCO-style bbox mAP against hand-worked cases and direct NumPy restatements — the published definitions of mmcls
BaseDataset.evaluate, mmseg eval_metrics and pycocotools COCOeval (none of the three packages is installed here)."""
import json
import os

import numpy as np
import pytest


def test_topk_accuracy():
    """Hand-worked: four samples, six classes."""
    from rscotr_amd.metrics import accuracy
    s = np.array([[0.1, 0.7, 0.2, 0.0, 0.0, 0.0], [0.5, 0.1, 0.1, 0.1, 0.1, 0.1], [0.0, 0.1, 0.2, 0.3, 0.25, 0.15], [0.3, 0.2, 0.1, 0.15, 0.15, 0.1]])
    # top-1 classes are 1, 0, 3, 0
    assert accuracy(list(s), [1, 1, 5, 2], topk=(1,))['accuracy_top-1'] == pytest.approx(25.0)
    assert accuracy(list(s), [1, 0, 5, 2], topk=(1,))['accuracy_top-1'] == pytest.approx(50.0)
    # top-2 sets: {1, 2}, {0, x}, {3, 4}, {0, 1}
    assert accuracy(list(s), [2, 0, 4, 1], topk=(2,))['accuracy_top-2'] == pytest.approx(100.0)
    assert accuracy(list(s), [2, 0, 4, 5], topk=(1, 2)) == {'accuracy_top-1': pytest.approx(25.0), 'accuracy_top-2': pytest.approx(75.0)}
    # a score threshold on top of the rank (mmcls `thrs`): sample 2's best score 0.3 does not exceed 0.4
    assert accuracy(list(s), [1, 0, 3, 0], topk=(1,), thrs=0.4)['accuracy_top-1'] == pytest.approx(50.0)


def test_topk_accuracy_exact():
    from rscotr_amd.metrics import accuracy
    rng = np.random.RandomState(0)
    s = rng.rand(200, 45).astype(np.float32)
    gt = rng.randint(0, 45, size=200)
    out = accuracy(list(s), gt, topk=(1, 5))
    order = np.argsort(-s, axis=1)
    assert out['accuracy_top-1'] == pytest.approx(100.0 * np.mean(order[:, 0] == gt))
    assert out['accuracy_top-5'] == pytest.approx(100.0 * np.mean((order[:, :5] == gt[:, None]).any(1)))


def _np_seg_metrics(preds, gts, C):
    inter, union, npred, nlab = np.zeros(C), np.zeros(C), np.zeros(C), np.zeros(C)
    for p, g in zip(preds, gts):
        g = g.astype(np.int64).copy()
        g[g == 0] = 255
        g = g - 1
        g[g == 254] = 255
        m = g != 255
        p, g = p[m], g[m]
        for c in range(C):
            inter[c] += np.sum((p == c) & (g == c)); npred[c] += np.sum(p == c); nlab[c] += np.sum(g == c)
    union = npred + nlab - inter
    return inter, union, npred, nlab


def test_seg_metrics_against_numpy(tmp_path):
    from PIL import Image
    from rscotr_amd.pipeline import TileSegDataset
    rng = np.random.RandomState(1)
    img_dir, ann_dir = tmp_path / 'img', tmp_path / 'ann'
    img_dir.mkdir(); ann_dir.mkdir()
    gts, preds = [], []
    for i in range(4):
        g = rng.randint(0, 7, size=(40, 52)).astype(np.uint8)  # raw labels 0 (ignored after reduce_zero_label) .. 6
        if i == 0:
            g[:5] = 255
        Image.fromarray(rng.randint(0, 255, size=(40, 52, 3)).astype(np.uint8)).save(img_dir / f't{i}.png')
        Image.fromarray(g).save(ann_dir / f't{i}.png')
        p = np.where(rng.rand(40, 52) < 0.6, np.clip(g.astype(np.int64) - 1, 0, 5), rng.randint(0, 6, size=(40, 52)))
        gts.append(g); preds.append(p)
    ds = TileSegDataset(str(img_dir), str(ann_dir))
    out = ds.evaluate(preds, metric=['mFscore', 'mIoU'], pre_eval=True, classwise=True, device='cpu')
    inter, union, npred, nlab = _np_seg_metrics(preds, gts, 6)
    r2 = lambda x: float(np.round(x * 100, 2)) / 100.0
    assert out['aAcc'] == r2(inter.sum() / nlab.sum())
    assert out['mIoU'] == r2(np.nanmean(inter / union)) and out['mAcc'] == r2(np.nanmean(inter / nlab))
    prec, rec = inter / npred, inter / nlab
    assert out['mFscore'] == r2(np.nanmean(2 * prec * rec / (prec + rec)))
    assert out['mPrecision'] == r2(np.nanmean(prec)) and out['mRecall'] == r2(np.nanmean(rec))
    for c, name in enumerate(ds.CLASSES):
        assert out[f'IoU.{name}'] == r2(inter[c] / union[c])
    with pytest.raises(KeyError):
        ds.evaluate(preds, metric='mAP', device='cpu')


def test_coco_map_hand_worked_case():
    """One class, one image, two ground truths, three detections: TP (0.9), FP (0.8), TP (0.7): precision envelope 1, 2/3, 2/3 at
    recalls 0.5, 0.5, 1.0 -> AP = (51 * 1 + 50 * 2/3) / 101 at every threshold the matches survive."""
    from rscotr_amd.metrics import coco_bbox_map
    gt = np.array([[10, 10, 60, 60], [100, 100, 180, 180]], dtype=np.float32)
    det = np.array([[10, 10, 60, 60, 0.9], [300, 300, 340, 340, 0.8], [100, 100, 180, 180, 0.7]], dtype=np.float32)
    out = coco_bbox_map([[det]], [gt], [np.array([0, 0])], ('a',), iou_thrs=[0.5])
    ap = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101
    assert out['bbox_mAP'] == float(f'{ap:.3f}') and out['bbox_mAP_50'] == float(f'{ap:.3f}')
    assert out['bbox_mAP_75'] == -1.0  # 0.75 is not among the thresholds
    # area ranges: gt 0 is 50 x 50 = medium, gt 1 is 80 x 80 = medium; no small / large ground truth
    assert out['bbox_mAP_s'] == -1.0 and out['bbox_mAP_l'] == -1.0 and out['bbox_mAP_m'] == float(f'{ap:.3f}')
    # default thresholds .5:.05:.95 with exact boxes: the same AP at all ten
    assert coco_bbox_map([[det]], [gt], [np.array([0, 0])], ('a',))['bbox_mAP'] == float(f'{ap:.3f}')


def test_coco_map_properties():
    from rscotr_amd.metrics import coco_bbox_map
    rng = np.random.RandomState(3)
    K, n_img = 4, 6
    gtb, gtl, perfect, shifted = [], [], [], []
    for _ in range(n_img):
        m = rng.randint(1, 6)
        xy = rng.rand(m, 2) * 300
        wh = 20 + rng.rand(m, 2) * 150
        b = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        l = rng.randint(0, K, size=m)
        gtb.append(b); gtl.append(l)
        perfect.append([np.concatenate([b[l == k], 0.5 + 0.5 * rng.rand(int((l == k).sum()), 1)], 1) for k in range(K)])
        sh = b + np.array([0.2, 0.2, 0.2, 0.2]) * np.concatenate([wh, wh], 1)  # IoU = 0.8^2 / (2 - 0.8^2) = 0.47 < 0.5
        shifted.append([np.concatenate([sh[l == k], rng.rand(int((l == k).sum()), 1)], 1) for k in range(K)])
    names = tuple('abcd')
    p = coco_bbox_map(perfect, gtb, gtl, names, classwise=True)
    assert p['bbox_mAP'] == 1.0 and p['bbox_mAP_50'] == 1.0 and p['bbox_mAP_75'] == 1.0
    assert all(p[f'bbox_AP.{n}'] == 1.0 for n in names if any((l == names.index(n)).any() for l in gtl))
    z = coco_bbox_map(shifted, gtb, gtl, names)
    assert z['bbox_mAP'] == 0.0
    # duplicates of the true positives with scores below EVERY true positive's (0.05 .. 0.1 against 0.5 .. 1) are false positives
    # behind full recall: AP unchanged
    dup = [[np.concatenate([d, d * np.array([1, 1, 1, 1, 0.1])]) for d in img] for img in perfect]
    assert coco_bbox_map(dup, gtb, gtl, names)['bbox_mAP'] == 1.0
    # ... with scores above every original's the copies match and the originals trail as false positives: unchanged again
    dup_hi = [[np.concatenate([d, d + np.array([0, 0, 0, 0, 1.0])]) for d in img] for img in perfect]
    assert coco_bbox_map(dup_hi, gtb, gtl, names)['bbox_mAP'] == 1.0
    # ... in between they interleave with true positives of other images: AP drops
    dup_mid = [[np.concatenate([d, d * np.array([1, 1, 1, 1, 0.9])]) for d in img] for img in perfect]
    assert 0.5 < coco_bbox_map(dup_mid, gtb, gtl, names)['bbox_mAP'] < 1.0
    # one wrong high-score box in one class lowers only that class
    bad = [[d.copy() for d in img] for img in perfect]
    bad[0][0] = np.concatenate([bad[0][0], np.array([[900, 900, 950, 950, 2.0]])])
    b = coco_bbox_map(bad, gtb, gtl, names, classwise=True)
    assert b['bbox_mAP'] < 1.0 and b['bbox_AP.b'] == p['bbox_AP.b']


def test_dataset_evaluate_cls_and_det(tmp_path):
    from PIL import Image
    from rscotr_amd.pipeline import CocoDetDataset, FolderClsDataset
    rng = np.random.RandomState(4)
    root = tmp_path / 'cls'
    for c in ('airplane', 'beach', 'forest'):
        (root / c).mkdir(parents=True)
        for i in range(3):
            Image.fromarray(rng.randint(0, 255, size=(32, 32, 3)).astype(np.uint8)).save(root / c / f'{i}.jpg')
    ds = FolderClsDataset(str(root))
    scores = [np.eye(3)[(l + (i % 3 == 0)) % 3] for i, (_, l) in enumerate(ds.items)]  # every third sample one class off
    out = ds.evaluate(scores, metric='accuracy', metric_options=dict(topk=(1, 2)))
    assert out['accuracy_top-1'] == pytest.approx(100.0 * 6 / 9)
    with pytest.raises(ValueError):
        ds.evaluate(scores, metric='f1_score')
    img_dir = tmp_path / 'det'
    img_dir.mkdir()
    images, anns = [], []
    for i in range(3):
        Image.fromarray(rng.randint(0, 255, size=(200, 240, 3)).astype(np.uint8)).save(img_dir / f'{i}.jpg')
        images.append(dict(id=i, file_name=f'{i}.jpg', width=240, height=200))
        for j in range(2):
            anns.append(dict(id=len(anns), image_id=i, category_id=1 + (i + j) % 2, bbox=[10 + 60 * j, 20, 50, 40 + 10 * i], area=50 * (40 + 10 * i), iscrowd=0))
    ann = tmp_path / 'ann.json'
    ann.write_text(json.dumps(dict(images=images, annotations=anns, categories=[dict(id=1, name='ship'), dict(id=2, name='dam')])))
    dd = CocoDetDataset(str(ann), str(img_dir), classes=('ship', 'dam'))
    res = [[np.concatenate([b[l == k], np.full((int((l == k).sum()), 1), 0.9)], 1) for k in range(2)] for _, b, l in dd.items]
    out = dd.evaluate(res, metric='bbox', iou_thrs=[0.5], classwise=True)
    assert out['bbox_mAP'] == 1.0 and out['bbox_mAP_50'] == 1.0 and out['bbox_AP.ship'] == 1.0 and isinstance(out['bbox_mAP_copypaste'], str)

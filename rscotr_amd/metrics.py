"""Task metrics behind `dataset.evaluate(results, **eval_kwargs)`: what the evaluation
hook of the reference — `mtl/runner/hooks/evaluation.py:127-148` — and `tools/test.py:89-222` get from the three un-vendored
dataset classes, restated from their published definitions:

* `accuracy(...)`           mmcls `BaseDataset.evaluate(metric='accuracy')`: top-k accuracy in percent (`accuracy_top-1`, `-5`).
* `seg_metrics(...)`        mmseg `CustomDataset.evaluate(metric=['mIoU', 'mFscore', 'mDice'])` / `eval_metrics`: one confusion
                            matrix accumulated ON THE DEVICE (`torch.bincount` of gt * C + pred over every image), then aAcc, per-class
                            IoU / Acc / Dice / F-score / precision / recall and their nan-means, as fractions rounded like mmseg's
                            (`round(x * 100, 2) / 100`).
* `coco_bbox_map(...)`      mmdet `CocoDataset.evaluate(metric='bbox', iou_thrs=..., classwise=...)` = pycocotools `COCOeval`
                            (bbox, no crowd): per (image, class) greedy matching in score order at every IoU threshold with the
                            area-range ignore rules, 101-point interpolated precision, AP = mean over thresholds x recall points x
                            classes; `bbox_mAP`, `_50`, `_75`, `_s`, `_m`, `_l` (-1 where undefined, 3 decimals) and the copy-paste
                            string.  Host NumPy: a few thousand boxes.

The reference's evaluation config (configs/multi/MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py:222-237) asks for
`accuracy`, `bbox` at `iou_thrs=[0.5]` classwise, and `['mFscore', 'mIoU']` classwise."""
from collections import OrderedDict

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------------------------- cls
def accuracy(results, gt_labels, topk=(1, 5), thrs=None):
    """results: per-sample score vectors; -> {'accuracy_top-k': percent}.  mmcls counts a sample as correct when the label is
    among its k highest scores (and, with `thrs`, that score exceeds the threshold)."""
    scores = torch.as_tensor(np.stack([np.asarray(r, dtype=np.float32) for r in results]))
    gt = torch.as_tensor(np.asarray(gt_labels, dtype=np.int64))
    assert scores.shape[0] == gt.shape[0], 'one score vector per sample'
    out = OrderedDict()
    maxk = min(max(topk), scores.shape[1])
    val, idx = scores.topk(maxk, dim=1)
    hit = idx == gt[:, None]
    for k in topk:
        kk = min(k, maxk)
        ok = hit[:, :kk]
        if thrs is not None:
            ok = ok & (val[:, :kk] > thrs)
        out[f'accuracy_top-{k}'] = float(ok.any(dim=1).float().mean() * 100.0)
    return out


# ---------------------------------------------------------------------------------------------------------------------- seg
def confusion_matrix(preds, gts, num_classes, ignore_index=255, reduce_zero_label=False, device=None):
    """Sum over images of the (gt, pred) histogram of the pixels whose label is not `ignore_index`.  gts: raw label maps
    (uint8); with `reduce_zero_label` label 0 becomes the ignore index and the others shift down by one (mmseg
    LoadAnnotations).  Accumulated on `device` (default: the GPU when there is one) in int64."""
    device = torch.device(device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu'))
    cm = torch.zeros(num_classes * num_classes, dtype=torch.int64, device=device)
    for p, g in zip(preds, gts):
        p = torch.as_tensor(np.asarray(p)).to(device).long().reshape(-1)
        g = torch.as_tensor(np.asarray(g)).to(device).long().reshape(-1)
        assert p.numel() == g.numel(), 'prediction and label map differ in size'
        if reduce_zero_label:
            g = torch.where(g == 0, torch.full_like(g, 255), g)
            g = torch.where(g == 255, g, g - 1)
            g = torch.where(g == 254, torch.full_like(g, 255), g)
        keep = (g != ignore_index) & (g < num_classes) & (p >= 0) & (p < num_classes)
        cm += torch.bincount(g[keep] * num_classes + p[keep], minlength=num_classes * num_classes)
    return cm.view(num_classes, num_classes)


def seg_metrics(cm, class_names, metrics=('mIoU',), beta=1, nan_to_num=None):
    """mmseg `total_area_to_metrics` + the key layout of `CustomDataset.evaluate` from a confusion matrix (rows = label)."""
    metrics = [metrics] if isinstance(metrics, str) else list(metrics)
    allowed = ('mIoU', 'mDice', 'mFscore')
    if not set(metrics).issubset(allowed):
        raise KeyError(f'metrics {metrics} is not supported')
    cm = cm.double().cpu()
    inter = cm.diag()
    label, pred = cm.sum(1), cm.sum(0)
    union = label + pred - inter
    ret = OrderedDict(aAcc=(inter.sum() / label.sum()).reshape(1))
    for m in metrics:
        if m == 'mIoU':
            ret['IoU'], ret['Acc'] = inter / union, inter / label
        elif m == 'mDice':
            ret['Dice'], ret['Acc'] = 2 * inter / (pred + label), inter / label
        else:
            prec, rec = inter / pred, inter / label
            ret['Fscore'] = (1 + beta ** 2) * prec * rec / ((beta ** 2) * prec + rec)
            ret['Precision'], ret['Recall'] = prec, rec
    ret = OrderedDict((k, v.numpy()) for k, v in ret.items())
    if nan_to_num is not None:
        ret = OrderedDict((k, np.nan_to_num(v, nan=nan_to_num)) for k, v in ret.items())
    out = OrderedDict()
    for k, v in ret.items():  # summary: aAcc, m<metric>
        out[k if k == 'aAcc' else 'm' + k] = float(np.round(np.nanmean(v) * 100, 2)) / 100.0
    for k, v in ret.items():  # per class
        if k != 'aAcc':
            for name, x in zip(class_names, v):
                out[f'{k}.{name}'] = float(np.round(x * 100, 2)) / 100.0
    return out


# ---------------------------------------------------------------------------------------------------------------------- det
_AREA = OrderedDict(all=(0.0, 1e10), small=(0.0, 32.0 ** 2), medium=(32.0 ** 2, 96.0 ** 2), large=(96.0 ** 2, 1e10))


def _iou_xyxy(d, g):
    """(len(d), len(g)) IoU of continuous boxes (pycocotools maskUtils.iou on xywh boxes, no crowd)."""
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    x1 = np.maximum(d[:, None, 0], g[None, :, 0])
    y1 = np.maximum(d[:, None, 1], g[None, :, 1])
    x2 = np.minimum(d[:, None, 2], g[None, :, 2])
    y2 = np.minimum(d[:, None, 3], g[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    ad = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1])
    ag = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
    return inter / (ad[:, None] + ag[None, :] - inter)


def _evaluate_img(dt, gt, rng, iou_thrs, max_det):
    """COCOeval.evaluateImg for one (image, class, area range): dt (n, 5) xyxy + score, gt (m, 4).  -> (dt scores, dt matched
    [T, n], dt ignored [T, n], number of non-ignored gts)."""
    g_area = (gt[:, 2] - gt[:, 0]) * (gt[:, 3] - gt[:, 1]) if len(gt) else np.zeros(0)
    g_ig = (g_area < rng[0]) | (g_area > rng[1])
    gi = np.argsort(g_ig, kind='mergesort')  # non-ignored first
    gt, g_ig = gt[gi], g_ig[gi]
    di = np.argsort(-dt[:, 4], kind='mergesort')[:max_det]
    dt = dt[di]
    T, D, G = len(iou_thrs), len(dt), len(gt)
    ious = _iou_xyxy(dt[:, :4], gt)
    gtm = -np.ones((T, G), dtype=np.int64)
    dtm = -np.ones((T, D), dtype=np.int64)
    dt_ig = np.zeros((T, D), dtype=bool)
    for ti, t in enumerate(iou_thrs):
        for d in range(D):
            iou, m = min(t, 1 - 1e-10), -1
            for g in range(G):
                if gtm[ti, g] >= 0:
                    continue  # (no crowd boxes: a matched gt is taken)
                if m > -1 and not g_ig[m] and g_ig[g]:
                    break  # a regular match exists and only ignored gts follow
                if ious[d, g] < iou:
                    continue
                iou, m = ious[d, g], g
            if m == -1:
                continue
            dt_ig[ti, d] = g_ig[m]
            dtm[ti, d], gtm[ti, m] = m, d
    d_area = (dt[:, 2] - dt[:, 0]) * (dt[:, 3] - dt[:, 1]) if D else np.zeros(0)
    out_rng = (d_area < rng[0]) | (d_area > rng[1])
    dt_ig = dt_ig | ((dtm < 0) & out_rng[None, :])
    return dt[:, 4], dtm >= 0, dt_ig, int((~g_ig).sum())


def coco_bbox_map(results, gt_boxes, gt_labels, class_names, iou_thrs=None, max_det=100, classwise=False):
    """results: per image a list (one entry per class) of (k, 5) arrays [x1, y1, x2, y2, score] in original-image coordinates;
    gt_boxes / gt_labels: per image (m, 4) xyxy and (m,) class indices.  -> the dict of mmdet CocoDataset.evaluate(metric='bbox')."""
    iou_thrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True) if iou_thrs is None else np.asarray(iou_thrs, dtype=np.float64)
    rec_thrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
    K, T, R = len(class_names), len(iou_thrs), len(rec_thrs)
    assert len(results) == len(gt_boxes) == len(gt_labels), 'one result per image'
    precision = -np.ones((T, R, K, len(_AREA)))
    for k in range(K):
        for a, rng in enumerate(_AREA.values()):
            scores, tps, igs, npig = [], [], [], 0
            for res, gb, gl in zip(results, gt_boxes, gt_labels):
                dt = np.asarray(res[k], dtype=np.float64).reshape(-1, 5)
                gt = np.asarray(gb, dtype=np.float64).reshape(-1, 4)[np.asarray(gl).reshape(-1) == k]
                if len(dt) == 0 and len(gt) == 0:
                    continue
                s, m, ig, n = _evaluate_img(dt, gt, rng, iou_thrs, max_det)
                scores.append(s); tps.append(m); igs.append(ig); npig += n
            if npig == 0:
                continue
            s = np.concatenate(scores) if scores else np.zeros(0)
            order = np.argsort(-s, kind='mergesort')
            m = np.concatenate(tps, axis=1)[:, order] if tps else np.zeros((T, 0), dtype=bool)
            ig = np.concatenate(igs, axis=1)[:, order] if igs else np.zeros((T, 0), dtype=bool)
            tp_sum = np.cumsum(m & ~ig, axis=1).astype(np.float64)
            fp_sum = np.cumsum(~m & ~ig, axis=1).astype(np.float64)
            for ti in range(T):
                tp, fp = tp_sum[ti], fp_sum[ti]
                rc = tp / npig
                pr = (tp / (fp + tp + np.spacing(1))).tolist()
                for i in range(len(pr) - 1, 0, -1):  # precision envelope
                    if pr[i] > pr[i - 1]:
                        pr[i - 1] = pr[i]
                inds = np.searchsorted(rc, rec_thrs, side='left')
                q = np.zeros(R)
                for ri, pi in enumerate(inds):
                    if pi < len(pr):
                        q[ri] = pr[pi]
                precision[ti, :, k, a] = q

    def summarize(iou=None, area='all'):
        s = precision[:, :, :, list(_AREA).index(area)]
        if iou is not None:
            s = s[np.where(np.isclose(iou_thrs, iou))[0]]
        s = s[s > -1]
        return float(np.mean(s)) if s.size else -1.0
    stats = [summarize(), summarize(.5), summarize(.75), summarize(area='small'), summarize(area='medium'), summarize(area='large')]
    out = OrderedDict()
    if classwise:  # mmdet logs the per-class AP table; the values ride along under the class names
        for k, name in enumerate(class_names):
            s = precision[:, :, k, 0]
            s = s[s > -1]
            out[f'bbox_AP.{name}'] = float(f'{np.mean(s):0.3f}') if s.size else float('nan')
    for key, v in zip(('mAP', 'mAP_50', 'mAP_75', 'mAP_s', 'mAP_m', 'mAP_l'), stats):
        out[f'bbox_{key}'] = float(f'{v:.3f}')
    out['bbox_mAP_copypaste'] = ' '.join(f'{v:.3f}' for v in stats)
    return out

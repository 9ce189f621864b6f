"""Device-side input path: the per-sample pipeline tail of the reference's dataset configs and
the collate step as ONE HIP launch per batch (`rscotr_img_prep_u8`, `rscotr_seg_label_prep_u8`), plus thin readers for
the three datasets' on-disk layouts.

Reference pipelines (configs/_base_/cls/resisc_swin_224.py:7-39, configs/_base_/det/dior.py:11-20,
configs/_base_/seg/potsdam_IRRG_all.py:8-19):
    decode -> [resize / RandAugment / PhotoMetricDistortion: host, not here] -> RandomCrop window (seg)
           -> RandomFlip -> Normalize(mean, std, to_rgb) -> Pad -> ImageToTensor / DefaultFormatBundle -> collate.
Everything from the crop window on runs on the device; the host only draws the random decisions (with the same NumPy
calls and in the same order as the mm* transforms, so a seeded run makes the same decisions) and uploads the raw bytes
through one pinned staging buffer.  There is no CPU fallback: without the HIP library `collate` raises.
"""
import ctypes
import json
import os

import numpy as np
import torch

from ._lib import lib
from . import ops

IMG_NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
META = 10


def _host_floats(v):
    arr = (ctypes.c_float * 3)(*[float(x) for x in v])
    return arr, ctypes.cast(arr, ctypes.c_void_p)


def _round_up(x, d):
    return (x + d - 1) // d * d


class DeviceCollate:
    """Batch builder for one task.  `__call__(samples, rng=None)` takes the decoded samples of one batch
    (dicts with `img`: HWC uint8 BGR ndarray, and per task `gt_label` | `gt_bboxes`, `gt_labels` | `gt_semantic_seg`:
    HW uint8) and returns the batch dict `MTL.train_step` consumes, tensors on `device`."""

    def __init__(self, task, device, img_norm_cfg=None, flip_prob=0.5, size_divisor=None, crop_size=None,
                 cat_max_ratio=1.0, reduce_zero_label=False, seg_pad_val=255, ignore_index=255):
        assert task in ('cls', 'det', 'seg')
        self.task, self.device = task, torch.device(device)
        cfg = dict(IMG_NORM if img_norm_cfg is None else img_norm_cfg)
        self.mean, self.std, self.to_rgb = cfg['mean'], cfg['std'], bool(cfg.get('to_rgb', True))
        self.flip_prob, self.size_divisor, self.crop_size = flip_prob, size_divisor, crop_size
        self.cat_max_ratio, self.reduce_zero_label = cat_max_ratio, reduce_zero_label
        self.seg_pad_val, self.ignore_index = seg_pad_val, ignore_index
        self._stage = None  # pinned byte staging buffer (grow-only)

    # ---- host-side random decisions (the draws of mmcv / mmseg / mmdet RandomFlip and mmseg RandomCrop) ----------
    def _crop_window(self, img, seg, rng):
        """mmseg RandomCrop.get_crop_bbox + the cat_max_ratio retry loop (up to 10 draws)."""
        H, W = img.shape[:2]
        ch, cw = self.crop_size

        def draw():
            my, mx = max(H - ch, 0), max(W - cw, 0)
            oy, ox = rng.randint(0, my + 1), rng.randint(0, mx + 1)
            return ox, oy, min(cw, W - ox), min(ch, H - oy)
        win = draw()
        if self.cat_max_ratio < 1.0 and seg is not None:
            for _ in range(10):
                x0, y0, w, h = win
                lab, cnt = np.unique(seg[y0:y0 + h, x0:x0 + w], return_counts=True)
                # the reference counts on the label map AFTER LoadAnnotations: with reduce_zero_label the raw values 0
                # and 255 are both the ignore index there
                keep = ((lab != 0) & (lab != 255)) if self.reduce_zero_label else (lab != self.ignore_index)
                cnt = cnt[keep]
                if len(cnt) > 1 and cnt.max() / cnt.sum() < self.cat_max_ratio:
                    break
                win = draw()
        return win

    def _stage_bytes(self, arrays):
        total = sum(a.nbytes for a in arrays)
        if self._stage is None or self._stage.numel() < total:
            self._stage = torch.empty(max(total, 1 << 20), dtype=torch.uint8,
                                      pin_memory=self.device.type == 'cuda')
        offs, o = [], 0
        view = self._stage.numpy()
        for a in arrays:
            view[o:o + a.nbytes] = np.ascontiguousarray(a).reshape(-1)
            offs.append(o)
            o += a.nbytes
        return self._stage[:total].to(self.device, non_blocking=True), offs

    def __call__(self, samples, rng=None):
        rng = rng or np.random
        B = len(samples)
        imgs = [s['img'] for s in samples]
        segs = [s.get('gt_semantic_seg') for s in samples]
        wins, flips = [], []
        for img, seg in zip(imgs, segs):
            assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3, 'decoded HWC uint8 images expected'
            H, W = img.shape[:2]
            wins.append(self._crop_window(img, seg, rng) if self.crop_size else (0, 0, W, H))
            flips.append(bool(rng.rand() < self.flip_prob))
        if self.crop_size:
            Hout, Wout = self.crop_size
        else:
            Hout, Wout = max(w[3] for w in wins), max(w[2] for w in wins)
            if self.size_divisor:
                Hout, Wout = _round_up(Hout, self.size_divisor), _round_up(Wout, self.size_divisor)
        buf, offs = self._stage_bytes(imgs)
        meta = torch.tensor([[o, im.shape[0], im.shape[1], im.shape[1] * 3, w[0], w[1], w[2], w[3], int(f), 0]
                             for o, im, w, f in zip(offs, imgs, wins, flips)], dtype=torch.int64).to(self.device)
        out = torch.empty((B, 3, Hout, Wout), dtype=torch.float32, device=self.device)
        mean_keep, mean_p = _host_floats(self.mean)
        std_keep, std_p = _host_floats(self.std)
        lib.call('rscotr_img_prep_u8', buf.data_ptr(), meta.data_ptr(), out.data_ptr(), B, Hout, Wout, mean_p, std_p,
                 int(self.to_rgb), ops._stream())
        metas = [dict(ori_shape=im.shape, img_shape=(w[3], w[2], 3), pad_shape=(Hout, Wout, 3), flip=f,
                      flip_direction='horizontal' if f else None, scale_factor=1.0,
                      img_norm_cfg=dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb))
                 for im, w, f in zip(imgs, wins, flips)]
        batch = dict(img=out, img_metas=metas)
        if self.task == 'cls':
            batch['gt_label'] = torch.tensor([int(s['gt_label']) for s in samples], dtype=torch.int64, device=self.device)
        elif self.task == 'det':
            boxes, labels, hboxes, hlabels = [], [], [], []
            for s, w, f in zip(samples, wins, flips):
                hb = np.asarray(s['gt_bboxes'], dtype=np.float32).reshape(-1, 4)
                if f:  # mmdet RandomFlip.bbox_flip, horizontal (on the host: the boxes are a few dozen floats)
                    hb = np.stack([np.float32(w[2]) - hb[:, 2], hb[:, 1], np.float32(w[2]) - hb[:, 0], hb[:, 3]], -1)
                hl = np.asarray(s['gt_labels'], dtype=np.int64).reshape(-1)
                boxes.append(torch.from_numpy(np.ascontiguousarray(hb)).to(self.device))
                labels.append(torch.from_numpy(np.ascontiguousarray(hl)).to(self.device))
                hboxes.append(np.ascontiguousarray(hb))
                hlabels.append(np.ascontiguousarray(hl))
            batch['gt_bboxes'], batch['gt_labels'] = boxes, labels
            batch['gt_bboxes_host'], batch['gt_labels_host'] = hboxes, hlabels  # (for the det head's packed batch layout)
        else:
            lbuf, loffs = self._stage_bytes(segs)
            lmeta = meta.clone()
            lmeta[:, 0] = torch.tensor(loffs, dtype=torch.int64)
            lmeta[:, 3] = torch.tensor([s.shape[1] for s in segs], dtype=torch.int64)
            lab = torch.empty((B, 1, Hout, Wout), dtype=torch.int64, device=self.device)
            lib.call('rscotr_seg_label_prep_u8', lbuf.data_ptr(), lmeta.data_ptr(), lab.data_ptr(), B, Hout, Wout,
                     int(self.reduce_zero_label), int(self.seg_pad_val), ops._stream())
            batch['gt_semantic_seg'] = lab
        return batch


def collate_for(task, device, **kw):
    """The three dataset configs' settings (configs/_base_/{cls/resisc_swin_224,det/dior,seg/potsdam_IRRG_all}.py)."""
    if task == 'cls':
        return DeviceCollate('cls', device, flip_prob=0.5, **kw)
    if task == 'det':
        return DeviceCollate('det', device, flip_prob=0.5, size_divisor=32, **kw)
    return DeviceCollate('seg', device, flip_prob=0.5, crop_size=(512, 512), cat_max_ratio=0.75, reduce_zero_label=True,
                         seg_pad_val=5, **kw)


# --------------------------------------------------------------------------------------------------------------
# on-disk layouts of the three datasets (decode with Pillow; BGR like mmcv.imread's default backend)
# --------------------------------------------------------------------------------------------------------------
def _imread_bgr(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGB'))[..., ::-1].copy()


class FolderClsDataset:
    """mmcls CustomDataset without an annotation file: `data_prefix/<class name>/<image>`; classes = sorted folder
    names (data/NWPU-RESISC45/train, configs/_base_/cls/resisc_swin_224.py:55-58)."""
    task = 'cls'
    EXT = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif')

    def __init__(self, data_prefix):
        self.CLASSES = sorted(d for d in os.listdir(data_prefix) if os.path.isdir(os.path.join(data_prefix, d)))
        self.items = [(os.path.join(data_prefix, c, f), i) for i, c in enumerate(self.CLASSES)
                      for f in sorted(os.listdir(os.path.join(data_prefix, c))) if f.lower().endswith(self.EXT)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        path, label = self.items[i]
        return dict(img=_imread_bgr(path), gt_label=label, filename=path)

    def evaluate(self, results, metric='accuracy', metric_options=None, indices=None, logger=None, **kwargs):
        """mmcls BaseDataset.evaluate: `results` = per-sample score vectors in dataset order -> {'accuracy_top-1', 'accuracy_top-5'}
        in percent (metric_options: topk, thrs)."""
        from .metrics import accuracy
        metrics = [metric] if isinstance(metric, str) else list(metric)
        if set(metrics) - {'accuracy'}:
            raise ValueError(f'metric {set(metrics) - {"accuracy"}} is not supported (accuracy only)')
        opt = dict(topk=(1, 5)) if metric_options is None else dict(metric_options)
        labels = [self.items[i][1] for i in (range(len(self.items)) if indices is None else indices)]
        assert len(results) == len(labels), 'dataset testing results should be of the same length as gt_labels'
        topk = opt.get('topk', (1, 5))
        return accuracy(results, labels, topk=(topk,) if isinstance(topk, int) else tuple(topk), thrs=opt.get('thrs'))


class CocoDetDataset:
    """mmdet CocoDataset on DIOR's converted annotations (configs/_base_/det/dior.py:41-47): images without boxes and
    crowd / degenerate boxes are dropped as mmdet's `_filter_imgs` / `_parse_ann_info` do; labels index `classes`."""
    task = 'det'

    def __init__(self, ann_file, img_prefix, classes):
        with open(ann_file) as fh:
            coco = json.load(fh)
        self.CLASSES = tuple(classes)
        cat = {c['id']: self.CLASSES.index(c['name']) for c in coco['categories'] if c['name'] in self.CLASSES}
        anns = {}
        for a in coco['annotations']:
            anns.setdefault(a['image_id'], []).append(a)
        self.items = []
        for im in coco['images']:
            boxes, labels = [], []
            for a in anns.get(im['id'], []):
                x, y, w, h = a['bbox']
                if a.get('ignore', False) or a.get('iscrowd', False) or a['category_id'] not in cat:
                    continue
                if w < 1 or h < 1 or a.get('area', w * h) <= 0:
                    continue
                if max(0, min(x + w, im['width']) - max(x, 0)) * max(0, min(y + h, im['height']) - max(y, 0)) == 0:
                    continue
                boxes.append([x, y, x + w, y + h])
                labels.append(cat[a['category_id']])
            if boxes and min(im['width'], im['height']) >= 32:
                self.items.append((os.path.join(img_prefix, im['file_name']), np.asarray(boxes, np.float32),
                                   np.asarray(labels, np.int64)))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        path, boxes, labels = self.items[i]
        return dict(img=_imread_bgr(path), gt_bboxes=boxes, gt_labels=labels, filename=path)

    def evaluate(self, results, metric='bbox', logger=None, jsonfile_prefix=None, classwise=False, proposal_nums=(100, 300, 1000),
                 iou_thrs=None, metric_items=None, **kwargs):
        """mmdet CocoDataset.evaluate(metric='bbox'): `results` = per image a list (per class) of (k, 5) arrays in original-image
        coordinates -> bbox_mAP / _50 / _75 / _s / _m / _l (COCOeval semantics, rscotr_amd/metrics.py)."""
        from .metrics import coco_bbox_map
        metrics = [metric] if isinstance(metric, str) else list(metric)
        if metrics != ['bbox']:
            raise KeyError(f'metric {metrics} is not supported (bbox only)')
        assert len(results) == len(self.items), 'one result per image'
        out = coco_bbox_map(results, [it[1] for it in self.items], [it[2] for it in self.items], self.CLASSES, iou_thrs=iou_thrs,
                            max_det=proposal_nums[0], classwise=classwise)
        if metric_items is not None:
            keep = {f'bbox_{m}' for m in metric_items}
            out = type(out)((k, v) for k, v in out.items() if k in keep or k == 'bbox_mAP_copypaste' or k.startswith('bbox_AP.'))
        return out


class TileSegDataset:
    """mmseg CustomDataset / PotsdamDataset: `img_dir/<name>.png` + `ann_dir/<name>.png` single-channel label tiles
    (configs/_base_/seg/potsdam_IRRG_all.py:52-62)."""
    task = 'seg'
    CLASSES = ('impervious_surface', 'building', 'low_vegetation', 'tree', 'car', 'clutter')

    def __init__(self, img_dir, ann_dir, img_suffix='.png', seg_map_suffix='.png', reduce_zero_label=True, ignore_index=255):
        names = sorted(f[:-len(img_suffix)] for f in os.listdir(img_dir) if f.endswith(img_suffix))
        self.items = [(os.path.join(img_dir, n + img_suffix), os.path.join(ann_dir, n + seg_map_suffix)) for n in names]
        self.reduce_zero_label, self.ignore_index = reduce_zero_label, ignore_index  # (mmseg PotsdamDataset: True, 255)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        from PIL import Image
        ip, ap = self.items[i]
        with Image.open(ap) as im:
            seg = np.asarray(im).astype(np.uint8)
        return dict(img=_imread_bgr(ip), gt_semantic_seg=seg, filename=ip)

    def evaluate(self, results, metric='mIoU', logger=None, gt_seg_maps=None, device=None, **kwargs):
        """mmseg CustomDataset.evaluate: `results` = per-image label maps at the original size (dataset order) -> aAcc, mIoU / mAcc,
        mFscore / mPrecision / mRecall, mDice and the per-class values, as fractions.  The confusion matrix is accumulated on the
        device (rscotr_amd/metrics.py); `pre_eval` / `classwise` of the reference's config are accepted and change nothing here
        (per-class values are always returned)."""
        from PIL import Image
        from .metrics import confusion_matrix, seg_metrics
        assert len(results) == len(self.items), 'one label map per image'

        def gts():
            if gt_seg_maps is not None:
                yield from gt_seg_maps
                return
            for _, ap in self.items:
                with Image.open(ap) as im:
                    yield np.asarray(im).astype(np.uint8)
        cm = confusion_matrix(results, gts(), len(self.CLASSES), ignore_index=self.ignore_index,
                              reduce_zero_label=self.reduce_zero_label, device=device)
        return seg_metrics(cm, self.CLASSES, metrics=metric)


class DeviceLoader:
    """Minimal batch loader over one of the datasets above: shuffled index batches, decoded on the host, everything
    else in `DeviceCollate`.  `len()` = batches per epoch; `.dataset.task` is what MultiDataLoader tags batches with."""

    def __init__(self, dataset, collate, batch_size, shuffle=True, drop_last=True, seed=0, test_mode=False):
        self.dataset, self.collate, self.batch_size = dataset, collate, batch_size
        self.shuffle, self.drop_last, self.rng = shuffle, drop_last, np.random.RandomState(seed)
        # test_mode: dataset order, every sample, batches of {task, img, img_metas} only — what `engine.single_gpu_test` feeds
        # `model(return_loss=False, **data)` (the collate should be built with flip_prob = 0 and no crop)
        self.test_mode = test_mode
        if test_mode:
            self.shuffle, self.drop_last = False, False

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = self.rng.permutation(len(self.dataset)) if self.shuffle else np.arange(len(self.dataset))
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            batch = self.collate([self.dataset[int(i)] for i in idx], self.rng)
            yield dict(task=self.dataset.task, img=batch['img'], img_metas=batch['img_metas']) if self.test_mode else batch

"""Classification decoders (`type='SlvlClsHead'`, `type='MlvlClsHead'` + `MlvlClsPixelDecoder`) and batch augments.

Mirrors models/multi/cls_head/slvl_cls_head.py:9-27 (mmcls LinearClsHead + GlobalAveragePooling
+ LabelSmoothLoss(0.1, 'original')), models/multi/cls_head/mlvl_cls_head.py:12-119 with
models/multi/cls_head/pixel_decoder.py:14-117 (the multi-level head of the `MTL_swin-t-...` configs: the four
neck maps go through the shared encoder, one of eight pooling schemes turns the memories into a 256-d token)
and the mmcls `Augments` (BatchMixup / BatchCutMix) that models/multi/multitask_learner.py:60-62,120-121
applies before the backbone.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .registry import MODELS


@MODELS.register_module()
class LabelSmoothLoss(nn.Module):
    def __init__(self, label_smooth_val, num_classes=None, mode='original', reduction='mean', loss_weight=1.0):
        super().__init__()
        assert mode == 'original' and reduction == 'mean'
        self.label_smooth_val, self.num_classes, self.loss_weight = label_smooth_val, num_classes, loss_weight

    def forward(self, cls_score, label, avg_factor=None):
        """label: (B,) int64 or (B,C) soft/one-hot labels."""
        C = cls_score.shape[-1]
        if label.dim() == 1:
            label = torch.nn.functional.one_hot(label, C).to(cls_score.dtype)
        return self.loss_weight * ops.soft_ce_label_smooth(cls_score, label, self.label_smooth_val,
                                                           float(avg_factor or cls_score.shape[0]))


@MODELS.register_module()
class SlvlClsHead(nn.Module):
    def __init__(self, num_classes, in_channels, loss=dict(type='CrossEntropyLoss', loss_weight=1.0),
                 cal_acc=False, topk=(1,), init_cfg=None):
        super().__init__()
        self.num_classes, self.in_channels, self.cal_acc = num_classes, in_channels, cal_acc
        self.compute_loss = MODELS.build(loss)
        self.fc = nn.Linear(in_channels, num_classes)
        # mmcls LinearClsHead default; mmcv applies an init_cfg entry to EVERY matching layer inside the module
        self.init_cfg = init_cfg if init_cfg is not None else dict(type='Normal', layer='Linear', std=0.01)

    def init_weights(self):
        for cfg in (self.init_cfg if isinstance(self.init_cfg, (list, tuple)) else [self.init_cfg]):
            layers = cfg.get('layer', ())
            layers = [layers] if isinstance(layers, str) else list(layers)
            for m in self.modules():
                if type(m).__name__ not in layers:
                    continue
                if cfg['type'] == 'Normal':
                    nn.init.normal_(m.weight, mean=cfg.get('mean', 0), std=cfg.get('std', 1))
                elif cfg['type'] == 'TruncNormal':
                    std = cfg.get('std', 1)
                    nn.init.trunc_normal_(m.weight, mean=cfg.get('mean', 0), std=std, a=cfg.get('a', -2), b=cfg.get('b', 2))
                elif cfg['type'] == 'Constant':
                    nn.init.constant_(m.weight, cfg['val'])
                else:
                    raise NotImplementedError(f"init_cfg type {cfg['type']}")
                if getattr(m, 'bias', None) is not None:
                    nn.init.constant_(m.bias, cfg.get('bias', 0))

    needs_neck = False  # the neck output is discarded by this head (slvl_cls_head.py:14-17): MTL skips computing it

    def pre_logits(self, x):
        return ops.global_avg_pool(x[-1])

    def _features(self, neck_feature, backbone_feature, shared_encoder):
        return self.pre_logits(backbone_feature)

    def forward_train(self, neck_feature, backbone_feature, gt_label, shared_encoder=None, **kwargs):
        cls_score = ops.linear(self._features(neck_feature, backbone_feature, shared_encoder), self.fc.weight, self.fc.bias, range_out=False)
        loss = self.compute_loss(cls_score, gt_label, avg_factor=len(cls_score))
        return dict(loss=loss)

    def simple_test(self, neck_feature, backbone_feature, shared_encoder=None, softmax=True, post_process=True):
        cls_score = ops.linear(self._features(neck_feature, backbone_feature, shared_encoder), self.fc.weight, self.fc.bias, range_out=False)
        pred = cls_score.softmax(-1) if softmax else cls_score
        return list(pred.detach().cpu().numpy()) if post_process else pred


@MODELS.register_module()
class MlvlClsPixelDecoder(nn.Module):
    """models/multi/cls_head/pixel_decoder.py:14-117: every neck level (low -> high resolution) + sine position +
    level embedding through the shared MSDeformAttn encoder; returns the per-level memories as (B, C, h, w) maps
    (channels-last views of the batch-first token tensor)."""

    def __init__(self, num_encoder_levels=4, strides=(4, 8, 16, 32), feat_channels=256, num_outs=4,
                 positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, normalize=True), init_cfg=None):
        super().__init__()
        self.strides = list(strides)
        self.num_encoder_levels = num_encoder_levels
        self.postional_encoding = MODELS.build(positional_encoding)  # (sic) the reference's attribute name
        self.level_encoding = nn.Embedding(num_encoder_levels, feat_channels)
        self.num_outs = num_outs

    def init_weights(self):
        nn.init.normal_(self.level_encoding.weight, mean=0, std=1)

    def forward(self, encoder, neck_feats):
        from .layers import LevelGeometry
        from .seg_head import _grid_refs
        n_in = len(neck_feats)
        B, device = neck_feats[0].shape[0], neck_feats[0].device
        inputs, poss, shapes, refs = [], [], [], []
        for i in range(self.num_encoder_levels):
            level_idx = n_in - i - 1
            f = neck_feats[level_idx]
            h, w = f.shape[-2:]
            pe = self.postional_encoding.unpadded(B, h, w, device)
            poss.append((self.level_encoding.weight[i].view(1, -1, 1, 1) + pe).flatten(2).transpose(1, 2))
            inputs.append(f.flatten(2).transpose(1, 2))
            shapes.append((h, w))
            refs.append(_grid_refs(h, w, self.strides[level_idx], device))
        geom = LevelGeometry.get(shapes, device)
        rkey = (tuple(shapes), B, str(device))   # a constant of the level shapes: dense once (as seg_head's pixel decoder)
        cache = self.__dict__.setdefault('_ref_cache', {})
        ref = cache.get(rkey)
        if ref is None:
            ref = cache[rkey] = torch.cat(refs, 0)[None, :, None].expand(B, -1, self.num_encoder_levels, -1).contiguous()
        memory = encoder(torch.cat(inputs, 1), None, None, query_pos=torch.cat(poss, 1), query_key_padding_mask=None,
                         reference_points=ref, **geom.kwargs())
        return [ops.tokens_to_map(memory[:, geom.starts[i]:geom.starts[i] + h * w], (h, w))
                for i, (h, w) in enumerate(shapes)]


@MODELS.register_module()
class MlvlClsHead(SlvlClsHead):
    """models/multi/cls_head/mlvl_cls_head.py:12-119.  `scheme` picks how the four encoder memories (low -> high
    resolution) become the classification token: 1 / 2 = average of level 0 / 1; 3 = average over all tokens;
    4 = mean of the per-level averages; 5 / 6 = learned weighting of the tokens of level 0 / 1 (a Linear over the
    token axis, built for 224x224 inputs: 4x4 / 7x7 tokens); 7 = the same over all levels; 8 = learned weighting
    of the per-level averages."""
    needs_neck = True
    _feat_length = {5: (4,), 6: (7,), 7: (4, 7, 14, 28)}

    def __init__(self, *args, pixel_decoder=None, scheme=5, **kwargs):
        super().__init__(*args, **kwargs)
        assert scheme in range(1, 9), 'scheme 0 is the reference\'s self-test mode'
        self.scheme = scheme
        self.pixel_decoder = pixel_decoder if isinstance(pixel_decoder, nn.Module) else MODELS.build(pixel_decoder)
        if scheme in self._feat_length:
            self.out_proj = nn.Linear(sum(x ** 2 for x in self._feat_length[scheme]), 1)
        elif scheme == 8:
            self.out_proj = nn.Linear(self.pixel_decoder.num_encoder_levels, 1)
        if hasattr(self, 'out_proj'):  # constant_init(out_proj, 1 / in_features) at construction (mlvl_cls_head.py:35,39)
            nn.init.constant_(self.out_proj.weight, 1.0 / self.out_proj.in_features)
            nn.init.constant_(self.out_proj.bias, 0)

    def init_weights(self):
        # as in the reference, the head's init_cfg then re-initialises every Linear in it (out_proj included)
        super().init_weights()
        self.pixel_decoder.init_weights()

    def pre_logits(self, mlvl_feats):
        s = self.scheme
        if s in (1, 2):
            return ops.global_avg_pool(mlvl_feats[s - 1])
        if s == 3:
            return torch.cat([f.flatten(2) for f in mlvl_feats], 2).mean(2)
        if s == 4:
            return sum(ops.global_avg_pool(f) for f in mlvl_feats) / len(mlvl_feats)
        if s in (5, 6):
            return self._token_proj(mlvl_feats[s - 5].flatten(2))
        if s == 7:
            return self._token_proj(torch.cat([f.flatten(2) for f in mlvl_feats], 2))
        return self._token_proj(torch.stack([ops.global_avg_pool(f) for f in mlvl_feats], -1))

    def _token_proj(self, seq):
        """nn.Linear(T, 1) over the last axis of (B, C, T), squeezed: a weighted token sum per channel."""
        B, C, T = seq.shape
        assert T == self.out_proj.in_features, f'scheme {self.scheme} is built for {self.out_proj.in_features} tokens, got {T}'
        return ops.linear(seq.reshape(B * C, T), self.out_proj.weight, self.out_proj.bias, range_out=False).view(B, C)

    def _features(self, neck_feature, backbone_feature, shared_encoder):
        return self.pre_logits(self.pixel_decoder(shared_encoder, neck_feature))


# ------------------------------------------------------------------------------------------
# mmcls Augments: one of the configured batch augments (or identity) per call
# ------------------------------------------------------------------------------------------
def _rand_bbox(img_h, img_w, lam, rng):
    """mmcls BatchCutMixLayer.rand_bbox + lam correction (correct_lam=True)."""
    ratio = np.sqrt(1 - lam)
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    cy = rng.randint(0, img_h)
    cx = rng.randint(0, img_w)
    yl = int(np.clip(cy - cut_h // 2, 0, img_h))
    yh = int(np.clip(cy + cut_h // 2, 0, img_h))
    xl = int(np.clip(cx - cut_w // 2, 0, img_w))
    xh = int(np.clip(cx + cut_w // 2, 0, img_w))
    lam = 1. - (yh - yl) * (xh - xl) / float(img_h * img_w)
    return (yl, yh, xl, xh), lam


class Augments:
    """cfg: list of dict(type='BatchMixup'|'BatchCutMix', alpha, num_classes, prob)."""

    def __init__(self, augments_cfg):
        if isinstance(augments_cfg, dict):
            augments_cfg = [augments_cfg]
        self.augs = [dict(c) for c in augments_cfg]
        self.probs = [a['prob'] for a in self.augs]
        assert sum(self.probs) <= 1.0 + 1e-12
        if 1 - sum(self.probs) > 0:
            self.augs.append(dict(type='Identity', num_classes=self.augs[0]['num_classes']))
            self.probs.append(1 - sum(self.probs[:len(self.augs) - 1]))
        self.num_classes = self.augs[0]['num_classes']

    def draw(self, batch_size, img_hw, rng=None):
        """Host-side random draw -> dict(kind, lam, index, bbox) (no device work)."""
        rng = rng or np.random
        aug = self.augs[rng.choice(len(self.augs), p=self.probs)]
        kind = {'BatchMixup': 'mixup', 'BatchCutMix': 'cutmix', 'Identity': 'identity'}[aug['type']]
        if kind == 'identity':
            return dict(kind=kind)
        lam = float(rng.beta(aug['alpha'], aug['alpha']))
        index = torch.from_numpy(rng.permutation(batch_size))
        d = dict(kind=kind, lam=lam, index=index)
        if kind == 'cutmix':
            d['bbox'], d['lam'] = _rand_bbox(img_hw[0], img_hw[1], lam, rng)
        return d

    # ---- shape-static formulation (hipGraph capture): every kind of draw becomes the same device
    # program, parameterised by three small tensors refreshed from the host draw before each replay.
    @staticmethod
    def static_params(draw, batch_size):
        """host draw -> dict(a (1,) f32, lam (1,) f32, index (B,) i64, box (4,) i64) on the CPU."""
        kind = draw['kind']
        lam = 1.0 if kind == 'identity' else float(draw['lam'])
        index = torch.arange(batch_size) if kind == 'identity' else draw['index'].long()
        box = torch.tensor(draw['bbox'] if kind == 'cutmix' else (0, 0, 0, 0), dtype=torch.long)
        a = lam if kind == 'mixup' else 1.0  # pixel blend outside the box
        return dict(a=torch.tensor([a], dtype=torch.float32), lam=torch.tensor([lam], dtype=torch.float32),
                    index=index, box=box)

    def apply_static(self, img, gt_label, sp):
        """mixup: a*img + (1-a)*img[idx]; cutmix: img[idx] inside the box, img outside (a = 1);
        identity: a = 1, idx = arange, empty box — bit-identical to __call__ for each kind."""
        onehot = torch.nn.functional.one_hot(gt_label, self.num_classes).to(img.dtype)
        H, W = img.shape[-2:]
        ys = torch.arange(H, device=img.device).view(H, 1)
        xs = torch.arange(W, device=img.device).view(1, W)
        box = sp['box']
        inside = (ys >= box[0]) & (ys < box[1]) & (xs >= box[2]) & (xs < box[3])
        other = img[sp['index']]
        a = sp['a']
        mixed = torch.where(inside, other, a * img + (1 - a) * other)
        lam = sp['lam']
        return mixed, lam * onehot + (1 - lam) * onehot[sp['index']]

    def __call__(self, img, gt_label, draw=None):
        if draw is None:
            draw = self.draw(img.shape[0], img.shape[-2:])
        onehot = torch.nn.functional.one_hot(gt_label, self.num_classes).to(img.dtype)
        if draw['kind'] == 'identity':
            return img, onehot
        idx = draw['index'].to(img.device)
        lam = draw['lam']
        if draw['kind'] == 'mixup':
            return lam * img + (1 - lam) * img[idx], lam * onehot + (1 - lam) * onehot[idx]
        y1, y2, x1, x2 = draw['bbox']
        img = img.clone()
        img[:, :, y1:y2, x1:x2] = img[idx, :, y1:y2, x1:x2]
        return img, lam * onehot + (1 - lam) * onehot[idx]

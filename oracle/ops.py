"""Oracle primitives (plain PyTorch, CPU). Test infrastructure only — see oracle/__init__.py."""
import math

import numpy as np
import torch
import torch.nn.functional as F

FP32_EPS = float(torch.finfo(torch.float32).eps)


# ----------------------------------------------------------------------------------------------
# MSDA sampling op — restates the mmcv-full 1.6.1 CUDA op `ms_deform_attn_forward`
# (ms_deform_attn_im2col_bilinear) that the reference reaches at
# models/multi/seg_head/pixel_decoder.py:134-146 and models/multi/bbox_head/transformer.py:211-221,
# 258-269.  Explicit 4-tap gather; autograd provides the backward.
# ----------------------------------------------------------------------------------------------
# Tests only (tests/parity.py): with BILINEAR_BAND = d (pixels) a sampling coordinate within d of a cell boundary is
# evaluated with the NEIGHBOURING cell's bilinear patch (floor moved by one): the sampled value is continuous across the
# boundary, its derivative with respect to the location is not — a coordinate that close to an integer is a coin toss
# between two correct fp32 implementations (x = loc * W - 0.5 carries ~1e-5 pixels of rounding at W ~ 100).  At the initial
# weights the DINO decoder samples exactly ON grid points (proposals at pixel centres, zero-initialised offsets).
BILINEAR_BAND = None


def _cell(x):
    """-> (x0, lw): lower grid index (float) and the weight of the upper one, with the band rule above."""
    x0 = torch.floor(x)
    if BILINEAR_BAND is not None:
        xd = x.detach()
        lw = xd - x0
        x0 = x0 - (lw < BILINEAR_BAND).to(x0.dtype) + (lw > 1 - BILINEAR_BAND).to(x0.dtype)
    return x0, x - x0


def msda_sample(value, spatial_shapes, level_start_index, loc, attn):
    """value (B,Nk,H,D); spatial_shapes [(H_l,W_l)]; loc (B,Nq,H,L,P,2) (x,y); attn (B,Nq,H,L,P)
    -> (B,Nq,H*D)."""
    B, Nk, H, D = value.shape
    _, Nq, _, L, P, _ = loc.shape
    shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if torch.is_tensor(spatial_shapes) else spatial_shapes)]
    starts = [int(s) for s in (level_start_index.tolist() if torch.is_tensor(level_start_index) else level_start_index)]
    out = value.new_zeros(B, Nq, H, D)
    bidx = torch.arange(B).view(B, 1, 1, 1).expand(B, Nq, H, P)
    hidx = torch.arange(H).view(1, 1, H, 1).expand(B, Nq, H, P)
    for l, (Hl, Wl) in enumerate(shapes):
        v = value[:, starts[l]:starts[l] + Hl * Wl]  # (B, Hl*Wl, H, D)
        x = loc[:, :, :, l, :, 0] * Wl - 0.5  # (B,Nq,H,P)
        y = loc[:, :, :, l, :, 1] * Hl - 0.5
        inside = (y > -1) & (x > -1) & (y < Hl) & (x < Wl)
        x0, lw = _cell(x)
        y0, lh = _cell(y)
        hw = 1 - lw
        hh = 1 - lh
        x0 = x0.long()
        y0 = y0.long()
        a = attn[:, :, :, l, :]
        samp = 0
        for (yy, xx, wgt) in ((y0, x0, hh * hw), (y0, x0 + 1, hh * lw),
                              (y0 + 1, x0, lh * hw), (y0 + 1, x0 + 1, lh * lw)):
            ok = inside & (yy >= 0) & (yy <= Hl - 1) & (xx >= 0) & (xx <= Wl - 1)
            idx = (yy.clamp(0, Hl - 1) * Wl + xx.clamp(0, Wl - 1))
            g = v[bidx, idx, hidx]  # (B,Nq,H,P,D)
            wg = torch.where(ok, wgt, torch.zeros_like(wgt))
            samp = samp + wg.unsqueeze(-1) * g
        out = out + (a.unsqueeze(-1) * samp).sum(3)
    return out.reshape(B, Nq, H * D)


def msda_sample_grid_sample(value, spatial_shapes, loc, attn):
    """Independent formulation through F.grid_sample (mmcv's `multi_scale_deformable_attn_pytorch`
    fallback) — used only to pin `msda_sample`."""
    B, _, H, D = value.shape
    _, Nq, _, L, P, _ = loc.shape
    shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if torch.is_tensor(spatial_shapes) else spatial_shapes)]
    vals = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * loc - 1
    sampled = []
    for l, (Hl, Wl) in enumerate(shapes):
        v = vals[l].flatten(2).transpose(1, 2).reshape(B * H, D, Hl, Wl)
        g = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False))
    a = attn.transpose(1, 2).reshape(B * H, 1, Nq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * a).sum(-1).view(B, H * D, Nq)
    return out.transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------------------------
# helpers restated from mmdet 2.25.1 (mmdet/models/utils/transformer.py, positional_encoding.py,
# core/bbox/transforms.py) — reached from models/multi/bbox_head/transformer.py and dino_head.py
# ----------------------------------------------------------------------------------------------
# Tests only (tests/parity.py): with RELU_BAND = d the gates whose pre-activation lies within d * mean|x| of zero are
# FLIPPED.  A ReLU gate that close to zero is a coin toss between any two correct fp32 implementations; evaluating the
# step once more with all of them flipped measures how far such flips can move each gradient tensor.
RELU_BAND = None


def relu(x):
    """F.relu (mmcv FFN / the branch MLPs: nn.ReLU)."""
    import torch.nn.functional as F
    if RELU_BAND is None:
        return F.relu(x)
    xd = x.detach()
    near = xd.abs() < RELU_BAND * xd.abs().mean()
    return x * ((xd > 0) ^ near).to(x.dtype)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


def sine_positional_encoding(mask, num_feats=128, temperature=10000, normalize=True,
                             scale=2 * math.pi, eps=1e-6, offset=0.0):
    """mmdet SinePositionalEncoding.forward; mask (B,H,W) bool (True = padded) -> (B,2*num_feats,H,W)."""
    mask = mask.to(torch.int)
    not_mask = 1 - mask
    y_embed = not_mask.cumsum(1, dtype=torch.get_default_dtype())
    x_embed = not_mask.cumsum(2, dtype=torch.get_default_dtype())
    if normalize:
        y_embed = (y_embed + offset) / (y_embed[:, -1:, :] + eps) * scale
        x_embed = (x_embed + offset) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.get_default_dtype())
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    B, H, W = mask.size()
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def bbox_cxcywh_to_xyxy(b):
    cx, cy, w, h = b.split((1, 1, 1, 1), dim=-1)
    return torch.cat([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def bbox_xyxy_to_cxcywh(b):
    x1, y1, x2, y2 = b.split((1, 1, 1, 1), dim=-1)
    return torch.cat([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], dim=-1)


def giou(b1, b2, aligned, eps=1e-6):
    """mmdet bbox_overlaps(mode='giou'); aligned: (n,4),(n,4)->(n,); else (n,4),(m,4)->(n,m)."""
    area1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    area2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    if aligned:
        lt = torch.max(b1[..., :2], b2[..., :2])
        rb = torch.min(b1[..., 2:], b2[..., 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1 + area2 - overlap
        elt = torch.min(b1[..., :2], b2[..., :2])
        erb = torch.max(b1[..., 2:], b2[..., 2:])
    else:
        lt = torch.max(b1[..., :, None, :2], b2[..., None, :, :2])
        rb = torch.min(b1[..., :, None, 2:], b2[..., None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1[..., None] + area2[..., None, :] - overlap
        elt = torch.min(b1[..., :, None, :2], b2[..., None, :, :2])
        erb = torch.max(b1[..., :, None, 2:], b2[..., None, :, 2:])
    e = union.new_tensor([eps])
    union = torch.max(union, e)
    ious = overlap / union
    ewh = (erb - elt).clamp(min=0)
    earea = torch.max(ewh[..., 0] * ewh[..., 1], e)
    return ious - (earea - union) / earea


# ----------------------------------------------------------------------------------------------
# losses — mmdet 2.25.1 FocalLoss (mmcv sigmoid_focal_loss CUDA op), L1Loss, GIoULoss, with
# weight_reduce_loss' `sum / (avg_factor + eps)`; reached from
# models/multi/bbox_head/mmdet_detr_head/detr_head.py:384-415 and dino_head.py:272-309
# ----------------------------------------------------------------------------------------------
def sigmoid_focal_loss_sum(pred, target, gamma=2.0, alpha=0.25):
    """pred (N,C) logits, target (N,) int64 in [0,C] (C = background). Returns the SUM."""
    N, C = pred.shape
    p = torch.sigmoid(pred)
    onehot = torch.zeros_like(pred)
    fg = target < C
    onehot[fg, target[fg]] = 1.0
    flt_min = torch.finfo(torch.float32).tiny
    term_p = (1 - p).pow(gamma) * torch.log(p.clamp(min=flt_min))
    term_n = p.pow(gamma) * torch.log((1 - p).clamp(min=flt_min))
    loss = -onehot * alpha * term_p - (1 - onehot) * (1 - alpha) * term_n
    return loss.sum()


def l1_loss_sum(pred, target, weight):
    return ((pred - target).abs() * weight).sum()


def giou_loss_sum(pred_xyxy, target_xyxy, weight4, eps=1e-6):
    w = weight4.mean(-1)
    return ((1 - giou(pred_xyxy, target_xyxy, aligned=True, eps=eps)) * w).sum()


# ----------------------------------------------------------------------------------------------
# Hungarian matching — mmdet 2.25.1 HungarianAssigner.assign + FocalLossCost/BBoxL1Cost/IoUCost
# (cfg train_cfg.det.assigner, configs/multi/MTL_slvlcls_...potsdam.py:169-174), called from
# models/multi/bbox_head/mmdet_detr_head/detr_head.py:513-515.  The solver is SciPy's
# linear_sum_assignment itself (the reference's real dependency, unpinned in requirement.txt).
# ----------------------------------------------------------------------------------------------
def match_cost(cls_score, bbox_pred, gt_bboxes, gt_labels, img_w, img_h,
               w_cls=2.0, w_l1=5.0, w_iou=2.0, alpha=0.25, gamma=2.0, eps=1e-12):
    factor = gt_bboxes.new_tensor([img_w, img_h, img_w, img_h]).unsqueeze(0)
    p = cls_score.sigmoid()
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    c_cls = (pos[:, gt_labels] - neg[:, gt_labels]) * w_cls
    gt_norm = gt_bboxes / factor
    c_l1 = torch.cdist(bbox_pred, bbox_xyxy_to_cxcywh(gt_norm), p=1) * w_l1
    boxes = bbox_cxcywh_to_xyxy(bbox_pred) * factor
    c_iou = -giou(boxes, gt_bboxes, aligned=False) * w_iou
    return c_cls + c_l1 + c_iou


def hungarian_assign(cost):
    """cost (Q,G) float32 tensor -> (pos_inds sorted int64, pos_assigned_gt_inds int64)."""
    from scipy.optimize import linear_sum_assignment
    Q, G = cost.shape
    if G == 0:
        return torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)
    r, c = linear_sum_assignment(cost.detach().cpu())
    assigned = torch.zeros(Q, dtype=torch.long)
    assigned[torch.from_numpy(r)] = torch.from_numpy(c) + 1
    pos = torch.nonzero(assigned > 0, as_tuple=False).squeeze(-1).unique()
    return pos, assigned[pos] - 1


# ----------------------------------------------------------------------------------------------
# nn.MultiheadAttention restated (torch 1.11 F.multi_head_attention_forward, dropout 0) as used
# by mmcv's MultiheadAttention wrapper (cfg ...potsdam.py:81-85, 144-151)
# ----------------------------------------------------------------------------------------------
def mha(query, key, value, in_w, in_b, out_w, out_b, num_heads, attn_mask=None):
    """query (Lq,B,C), key/value (Lk,B,C); attn_mask bool (Lq,Lk) or (B*heads,Lq,Lk), True = blocked."""
    Lq, B, C = query.shape
    Lk = key.shape[0]
    hd = C // num_heads
    q = F.linear(query, in_w[:C], in_b[:C])
    k = F.linear(key, in_w[C:2 * C], in_b[C:2 * C])
    v = F.linear(value, in_w[2 * C:], in_b[2 * C:])
    q = q.contiguous().view(Lq, B * num_heads, hd).transpose(0, 1)
    k = k.contiguous().view(Lk, B * num_heads, hd).transpose(0, 1)
    v = v.contiguous().view(Lk, B * num_heads, hd).transpose(0, 1)
    q = q * (float(hd) ** -0.5)
    s = torch.bmm(q, k.transpose(1, 2))
    if attn_mask is not None:
        m = attn_mask if attn_mask.dim() == 3 else attn_mask.unsqueeze(0)
        s = s.masked_fill(m, float('-inf'))
    a = torch.softmax(s, dim=-1)
    o = torch.bmm(a, v).transpose(0, 1).contiguous().view(Lq, B, C)
    return F.linear(o, out_w, out_b)

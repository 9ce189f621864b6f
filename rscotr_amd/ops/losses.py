"""Detection / segmentation loss pieces: box utilities, matching cost, the assignment solvers, focal / box loss sums, the
fused upsample + cross-entropy."""
import torch
from torch.autograd import Function

from .core import _WS, _Prof, _chk, _f32c, _ptr, _stream, lib

def bbox_cxcywh_to_xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def bbox_xyxy_to_cxcywh(b):
    x1, y1, x2, y2 = b.unbind(-1)
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], dim=-1)


def _giou(b1, b2, aligned, eps=1e-6):
    area1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    area2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    if aligned:
        lt, rb = torch.max(b1[..., :2], b2[..., :2]), torch.min(b1[..., 2:], b2[..., 2:])
        elt, erb = torch.min(b1[..., :2], b2[..., :2]), torch.max(b1[..., 2:], b2[..., 2:])
        a1, a2 = area1, area2
    else:
        lt = torch.max(b1[..., :, None, :2], b2[..., None, :, :2])
        rb = torch.min(b1[..., :, None, 2:], b2[..., None, :, 2:])
        elt = torch.min(b1[..., :, None, :2], b2[..., None, :, :2])
        erb = torch.max(b1[..., :, None, 2:], b2[..., None, :, 2:])
        a1, a2 = area1[..., None], area2[..., None, :]
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = (a1 + a2 - overlap).clamp(min=eps)
    ious = overlap / union
    ewh = (erb - elt).clamp(min=0)
    earea = (ewh[..., 0] * ewh[..., 1]).clamp(min=eps)
    return ious - (earea - union) / earea


def match_cost(cls_score, bbox_pred, gt_bboxes, gt_labels, img_w, img_h, w_cls, w_l1, w_iou, alpha, gamma, eps):
    """mmdet FocalLossCost + BBoxL1Cost(xywh) + IoUCost(giou) for S prediction sets of one image:
    cls_score (S,Q,C), bbox_pred (S,Q,4) cxcywh normalised, gt (G,4) xyxy pixels -> (S,Q,G)."""
    factor = gt_bboxes.new_tensor([img_w, img_h, img_w, img_h])
    p = cls_score.sigmoid()
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    c_cls = (pos[..., gt_labels] - neg[..., gt_labels]) * w_cls
    gt_c = bbox_xyxy_to_cxcywh(gt_bboxes / factor)
    c_l1 = (bbox_pred[..., :, None, :] - gt_c[None, None, :, :]).abs().sum(-1) * w_l1
    boxes = bbox_cxcywh_to_xyxy(bbox_pred) * factor
    c_iou = -_giou(boxes, gt_bboxes.unsqueeze(0).expand(boxes.shape[0], -1, -1), aligned=False) * w_iou
    return c_cls + c_l1 + c_iou


def match_cost_batched(cls_score, bbox_pred, gt_bboxes, gt_labels, factors, w_cls, w_l1, w_iou, alpha, gamma, eps):
    """mmdet FocalLossCost + BBoxL1Cost(xywh) + IoUCost(giou) for all images at once on padded ground truth, one
    kernel (rscotr_match_cost): cls_score (S,B,Q,C), bbox_pred (S,B,Q,4), gt_bboxes (B,G,4) xyxy pixels, gt_labels
    (B,G), factors (B,4) = (w,h,w,h) -> (S,B,Q,G).  Columns of padding ground truths hold finite garbage; the
    assignment ignores them."""
    cls_score, bbox_pred, gt_bboxes, factors = _f32c(cls_score), _f32c(bbox_pred), _f32c(gt_bboxes), _f32c(factors)
    gt_labels = gt_labels.contiguous()
    _chk(cls_score, bbox_pred, gt_bboxes, gt_labels, factors)
    S, B, Q, C = cls_score.shape
    G = gt_bboxes.shape[1]
    cost = torch.empty((S, B, Q, G), dtype=torch.float32, device=cls_score.device)
    lib.call('rscotr_match_cost', cls_score.data_ptr(), bbox_pred.data_ptr(), gt_bboxes.data_ptr(), gt_labels.data_ptr(),
             factors.data_ptr(), cost.data_ptr(), S, B, Q, C, G, float(w_cls), float(w_l1), float(w_iou), float(alpha),
             float(gamma), float(eps), _stream())
    return cost


def lsap_batch(flat_cost, rows, cols):
    """Solve len(rows) assignment problems whose fp32 costs are concatenated in `flat_cost`
    (device or host).  ONE device->host copy, then the C-ABI solver (rscotr_lsap_batch_f32).
    Returns (row_inds, col_inds): lists of int64 numpy arrays (row_inds ascending, as SciPy)."""
    import numpy as np
    host = flat_cost.detach().to('cpu', torch.float32).contiguous().numpy()  # the step's one sync
    n = len(rows)
    sizes = np.asarray(rows, dtype=np.int64) * np.asarray(cols, dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    outs = np.minimum(rows, cols).astype(np.int64)
    out_off = np.concatenate([[0], np.cumsum(outs)[:-1]]).astype(np.int64)
    total = int(outs.sum())
    r = np.zeros(max(total, 1), dtype=np.int64)
    c = np.zeros(max(total, 1), dtype=np.int64)
    rows_a = np.asarray(rows, dtype=np.int32)
    cols_a = np.asarray(cols, dtype=np.int32)
    assert host.size == int(sizes.sum())
    lib.call('rscotr_lsap_batch_f32', host.ctypes.data, offsets.ctypes.data, rows_a.ctypes.data,
             cols_a.ctypes.data, n, out_off.ctypes.data, r.ctypes.data, c.ctypes.data)
    return ([r[out_off[k]:out_off[k] + outs[k]] for k in range(n)],
            [c[out_off[k]:out_off[k] + outs[k]] for k in range(n)])


def lsap_device(cost, gcount):
    """The matcher's assignment problems solved ON THE DEVICE (rscotr_lsap_dev_f32: SciPy's algorithm and
    tie-breaks in fp64, one wavefront per problem, no host round trip).  cost (P, Q, ld) fp32 with the
    first gcount[p] columns of problem p real; gcount (P,) int32 device.  Returns q_for_gt (P, ld) int32:
    the query assigned to each ground truth, -1 for padding columns."""
    cost = _f32c(cost)
    _chk(cost, gcount)
    assert gcount.dtype == torch.int32 and gcount.is_contiguous()
    P, Q, ld = cost.shape
    out = torch.empty((P, ld), dtype=torch.int32, device=cost.device)
    lib.call('rscotr_lsap_dev_f32', cost.data_ptr(), gcount.data_ptr(), P, Q, ld, out.data_ptr(), _stream())
    return out


class _RefineBox(Function):
    @staticmethod
    def forward(ctx, delta, ref, eps):
        delta, ref = _f32c(delta), _f32c(ref)
        _chk(delta, ref)
        out = torch.empty_like(delta)
        lib.call('rscotr_refine_box_fwd', delta.data_ptr(), ref.data_ptr(), out.data_ptr(), delta.numel(), float(eps), _stream())
        ctx.save_for_backward(out, ref)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, g):
        out, ref = ctx.saved_tensors
        g = _f32c(g)
        dd = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        dr = torch.empty_like(out) if ctx.needs_input_grad[1] else None
        lib.call('rscotr_refine_box_bwd', g.data_ptr(), out.data_ptr(), ref.data_ptr(), _ptr(dd), _ptr(dr), out.numel(),
                 ctx.eps, _stream())
        return dd, dr, None


def refine_box(delta, ref, eps=1e-3):
    """sigmoid(delta + inverse_sigmoid(ref, eps)): one kernel per direction (rscotr_refine_box_*)."""
    return _RefineBox.apply(delta, ref, eps)


class _FocalSum(Function):
    @staticmethod
    def forward(ctx, pred, target, gamma, alpha, weight):
        pred = _f32c(pred)
        target = target.contiguous()
        weight = None if weight is None else _f32c(weight)
        _chk(pred, target, weight)
        S, N, C = pred.shape
        sums = torch.empty(S, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        lib.call('rscotr_focal_sum', pred.data_ptr(), target.data_ptr(), _ptr(weight), sums.data_ptr(), dpred.data_ptr(),
                 S, N, C, float(gamma), float(alpha), _stream())
        ctx.save_for_backward(dpred)
        return sums

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g.view(-1, 1, 1), None, None, None, None


def sigmoid_focal_loss_sum(pred, target, gamma, alpha, weight=None):
    """mmcv sigmoid_focal_loss (CUDA op semantics) summed per set, one kernel that also leaves the gradient
    (rscotr_focal_sum): pred (S,N,C) logits, target (S,N) int64 in [0,C] with C = background, optional per-sample
    weight (S,N) -> (S,)."""
    return _FocalSum.apply(pred, target, gamma, alpha, weight)


class _BoxLoss(Function):
    @staticmethod
    def forward(ctx, pred, target, weight, factors, eps):
        pred, target, weight, factors = _f32c(pred), _f32c(target), _f32c(weight), _f32c(factors)
        _chk(pred, target, weight, factors)
        S, B, Q, _ = pred.shape
        sums = torch.empty((2, S), dtype=torch.float32, device=pred.device)
        d_l1, d_gi = torch.empty_like(pred), torch.empty_like(pred)
        lib.call('rscotr_box_loss', pred.data_ptr(), target.data_ptr(), weight.data_ptr(), factors.data_ptr(),
                 sums.data_ptr(), d_l1.data_ptr(), d_gi.data_ptr(), S, B, Q, float(eps), _stream())
        ctx.save_for_backward(d_l1, d_gi)
        return sums[0], sums[1]

    @staticmethod
    def backward(ctx, g1, g2):
        d_l1, d_gi = ctx.saved_tensors
        return d_l1 * g1.view(-1, 1, 1, 1) + d_gi * g2.view(-1, 1, 1, 1), None, None, None, None


def box_loss_sums(pred, target, weight, factors, eps=1e-6):
    """L1 (cxcywh, normalised) and GIoU (xyxy in pixels: * factors (B,4)) loss sums per prediction set
    (detr_head.py:392-415), one kernel that also leaves both gradients (rscotr_box_loss):
    pred / target / weight (S,B,Q,4) -> (l1 (S,), giou (S,)); the GIoU weight is the mean of the 4 box weights."""
    return _BoxLoss.apply(pred, target, weight, factors, eps)


class _UpsampleCE(Function):
    @staticmethod
    def forward(ctx, logit, label, ignore_index):
        logit = _f32c(logit)
        label = label.contiguous()
        _chk(logit, label)
        B, C, h, w = logit.shape
        H, W = label.shape[-2:]
        lse = torch.empty((B, H, W), dtype=torch.float32, device=logit.device)
        sums = torch.empty(3, dtype=torch.float32, device=logit.device)
        with _Prof('upsample_ce_fwd', 4 * B * C * h * w + 12 * B * H * W):
            nws = lib.rscotr_upsample_ce_workspace()
            lib.call('rscotr_upsample_ce_fwd', logit.data_ptr(), label.data_ptr(), lse.data_ptr(), sums.data_ptr(),
                     B, C, h, w, H, W, int(ignore_index), _WS.get(nws, logit.device).data_ptr(), nws, _stream())
        ctx.save_for_backward(logit, label, lse)
        ctx.ignore = int(ignore_index)
        ctx.mark_non_differentiable(sums)
        npix = float(B * H * W)
        return sums[0] / npix, sums

    @staticmethod
    def backward(ctx, g_loss, g_sums):
        logit, label, lse = ctx.saved_tensors
        B, C, h, w = logit.shape
        H, W = label.shape[-2:]
        gscale = (g_loss / float(B * H * W)).reshape(1).float().contiguous()
        dlogit = torch.empty_like(logit)
        with _Prof('upsample_ce_bwd', 8 * B * C * h * w + 12 * B * H * W):
            lib.call('rscotr_upsample_ce_bwd', logit.data_ptr(), label.data_ptr(), lse.data_ptr(), gscale.data_ptr(),
                     dlogit.data_ptr(), B, C, h, w, H, W, ctx.ignore, _stream())
        return dlogit, None, None


def upsample_ce(seg_logit, label, ignore_index=255):
    """mmseg BaseDecodeHead.losses: bilinear resize (align_corners=False) to the label size, CE with
    ignore_index averaged over ALL pixels, and top-1 accuracy over non-ignored pixels — fused: the
    upsampled logits are never materialised (rscotr_upsample_ce_*).
    seg_logit (B,C,h,w), label (B,H,W) int64 -> (loss_ce 0-d, acc (1,))."""
    loss, sums = _UpsampleCE.apply(seg_logit, label, ignore_index)
    acc = (sums[1] * 100.0 / (sums[2] + torch.finfo(torch.float32).eps)).reshape(1)
    return loss, acc



class _DetProposals(Function):
    """Two-stage proposal selection of the DINO transformer (transformer.py:226-241) as one launch per direction
    (rscotr_det_proposals / _bwd): row maximum, top-k, the proposal add, both gathers and the sigmoid forward; the two
    scattered gradients backward (every row of d(enc_cls) / d(enc_reg) written once: no zero-fill, no scatter-add)."""

    @staticmethod
    def forward(ctx, enc_cls, enc_reg, proposals, K):
        B, N, C = enc_cls.shape
        cls2, reg2, prop = _f32c(enc_cls), _f32c(enc_reg), _f32c(proposals)
        assert reg2.shape == (B, N, 4) and prop.shape[1:] == (N, 4) and prop.shape[0] in (1, B)
        _chk(cls2, reg2, prop)
        dev = cls2.device
        idx = torch.empty((B, K), dtype=torch.int64, device=dev)
        score = torch.empty((B, K, C), dtype=torch.float32, device=dev)
        unact = torch.empty((B, K, 4), dtype=torch.float32, device=dev)
        anchor = torch.empty((B, K, 4), dtype=torch.float32, device=dev)
        inv = torch.empty((B, N), dtype=torch.int32, device=dev)
        lib.call('rscotr_det_proposals', cls2.data_ptr(), reg2.data_ptr(), prop.data_ptr(), int(prop.shape[0] == B and B > 1),
                 idx.data_ptr(), score.data_ptr(), unact.data_ptr(), anchor.data_ptr(), inv.data_ptr(), B, N, C, K, _stream())
        ctx.save_for_backward(anchor, inv)
        ctx.geom = (B, N, C, K)
        ctx.mark_non_differentiable(idx, unact)
        ctx.set_materialize_grads(False)
        return idx, score, unact, anchor

    @staticmethod
    def backward(ctx, _gi, g_score, _gu, g_anchor):
        anchor, inv = ctx.saved_tensors
        B, N, C, K = ctx.geom
        need = ctx.needs_input_grad
        gs = None if g_score is None else _f32c(g_score)
        ga = None if g_anchor is None else _f32c(g_anchor)
        d_cls = torch.empty((B, N, C), dtype=torch.float32, device=anchor.device) if need[0] else None
        d_reg = torch.empty((B, N, 4), dtype=torch.float32, device=anchor.device) if need[1] else None
        if d_cls is not None or d_reg is not None:
            lib.call('rscotr_det_proposals_bwd', _ptr(gs), _ptr(ga), anchor.data_ptr(), inv.data_ptr(), _ptr(d_cls), _ptr(d_reg),
                     B, N, C, K, _stream())
        return d_cls, d_reg, None, None


DET_PROPOSALS_MAX_N, DET_PROPOSALS_MAX_K = 36864, 1024


def det_proposals(enc_cls, enc_reg, proposals, K):
    """-> (topk_idx (B,K) int64, topk_score (B,K,C), topk_unact (B,K,4) detached, topk_anchor (B,K,4)): the K tokens with the
    largest class score (torch.topk order: descending; equal scores: lower index first), their class rows, enc_reg + proposals
    and its sigmoid.  enc_cls (B,N,C), enc_reg (B,N,4), proposals (B|1,N,4) (no gradient)."""
    return _DetProposals.apply(enc_cls, enc_reg, proposals.detach(), int(K))


def det_targets(q_for_gt, gt_lab, gt_boxn, Q, num_classes):
    """q_for_gt (S,B,G) int32 (ops.lsap_device), gt_lab (B,G) int64, gt_boxn (B,G,4) -> (labels (S,B,Q) int64, bbox_targets
    (S,B,Q,4), bbox_weights (S,B,Q,4)) of the assignment (detr_head.py:475-543), one launch."""
    S, B, G = q_for_gt.shape
    assert q_for_gt.dtype == torch.int32 and gt_lab.dtype == torch.int64 and gt_lab.shape == (B, G)
    q_for_gt, gt_lab, gt_boxn = q_for_gt.contiguous(), gt_lab.contiguous(), _f32c(gt_boxn)
    _chk(q_for_gt, gt_lab, gt_boxn)
    dev = gt_boxn.device
    labels = torch.empty((S, B, Q), dtype=torch.int64, device=dev)
    bt = torch.empty((S, B, Q, 4), dtype=torch.float32, device=dev)
    bw = torch.empty((S, B, Q, 4), dtype=torch.float32, device=dev)
    lib.call('rscotr_det_targets', q_for_gt.data_ptr(), gt_lab.data_ptr(), gt_boxn.data_ptr(), labels.data_ptr(), bt.data_ptr(),
             bw.data_ptr(), S, B, Q, G, int(num_classes), _stream())
    return labels, bt, bw

"""DINO detection decoder (`type='DINOHead'`) over the shared encoder.

Mirrors models/multi/bbox_head/{dino_head,transformer,query_denoising}.py and the vendored
mmdet_detr_head/{detr_head,deformable_detr_head}.py of the reference, with the same parameter
names.  Differences that do not change results:
  * tokens are batch-first (B,L,C);
  * target assignment for the 7 (interm + 6 decoder layers) x B matchings is batched: one
    device->host copy of all cost matrices, one call into the C-ABI LSAP solver, one host->device
    copy of the indices — instead of 7*B `cost.cpu()` round trips
    (mmdet_detr_head/detr_head.py:513-515);
  * normalisers (num_total_pos, cls_avg_factor) are known on the host from the GT counts
    (every GT is matched when num_query >= num_gt), so no `.item()` is needed; in distributed
    runs they are averaged across ranks exactly as `reduce_mean` does
    (mmdet_detr_head/detr_head.py:379-381,389-390; dino_head.py:266-268,282-283).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .layers import LevelGeometry, TransformerLayerSequence, inverse_sigmoid
from .registry import MODELS

FP32_EPS = float(torch.finfo(torch.float32).eps)


# ------------------------------------------------------------------------------------------
# loss / assigner config holders (mmdet FocalLoss, L1Loss, GIoULoss, HungarianAssigner + costs)
# ------------------------------------------------------------------------------------------
@MODELS.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0, activated=False):
        super().__init__()
        assert use_sigmoid and reduction == 'mean' and not activated
        self.use_sigmoid, self.gamma, self.alpha, self.loss_weight = use_sigmoid, gamma, alpha, loss_weight


@MODELS.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight


@MODELS.register_module()
class GIoULoss(nn.Module):
    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.loss_weight = eps, loss_weight


@MODELS.register_module()
class HungarianAssigner:
    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.), reg_cost=dict(type='BBoxL1Cost', weight=1.0),
                 iou_cost=dict(type='IoUCost', iou_mode='giou', weight=1.0)):
        assert cls_cost['type'] == 'FocalLossCost' and reg_cost['type'] == 'BBoxL1Cost' and iou_cost['type'] == 'IoUCost'
        assert reg_cost.get('box_format', 'xyxy') == 'xywh' and iou_cost.get('iou_mode', 'giou') == 'giou'
        self.w_cls, self.w_l1, self.w_iou = cls_cost.get('weight', 1.), reg_cost.get('weight', 1.), iou_cost.get('weight', 1.)
        self.alpha, self.gamma, self.eps = cls_cost.get('alpha', 0.25), cls_cost.get('gamma', 2.0), cls_cost.get('eps', 1e-12)


def build_mlp(input_dim, hidden_dim, output_dim, num_layers):
    h = [hidden_dim] * (num_layers - 1)
    layers = []
    for n, k in zip([input_dim] + h[:-1], h):
        layers.extend((nn.Linear(n, k), nn.ReLU()))
    layers.append(nn.Linear(hidden_dim, output_dim))
    return nn.Sequential(*layers)


def _mlp(x, seq):
    """Sequential(Linear, ReLU, ..., Linear) as one fused MLP (ops.mlp)."""
    return ops.mlp(x, [(m.weight, m.bias) for m in seq if isinstance(m, nn.Linear)], act='relu', range_out=False)  # (read by box arithmetic / losses)


# ------------------------------------------------------------------------------------------
# CDN query generator — query_denoising.py:8-201
# ------------------------------------------------------------------------------------------
class CdnQueryGenerator:
    def __init__(self, num_queries, hidden_dim, num_classes, noise_scale=dict(label=0.5, box=0.4),
                 group_cfg=dict(dynamic=True, num_groups=None, num_dn_queries=None)):
        self.num_queries, self.hidden_dim, self.num_classes = num_queries, hidden_dim, num_classes
        self.label_noise_scale, self.box_noise_scale = noise_scale['label'], noise_scale['box']
        self.dynamic_dn_groups = group_cfg.get('dynamic', False)
        self.num_dn = group_cfg['num_dn_queries'] if self.dynamic_dn_groups else group_cfg['num_groups']
        assert isinstance(self.num_dn, int) and self.num_dn >= 1

    def get_num_groups(self, group_queries=None):
        if self.dynamic_dn_groups:
            num_groups = 1 if group_queries == 0 else self.num_dn // group_queries
        else:
            num_groups = self.num_dn
        return max(int(num_groups), 1)

    def draw(self, total_gt, num_groups, device, generator=None):
        """The four random tensors the reference draws inline (query_denoising.py:116-144)."""
        K = 2 * num_groups * total_gt
        return dict(
            label_p=torch.rand(K, device=device, generator=generator),
            new_label=torch.randint(0, self.num_classes, (K,), device=device, generator=generator),
            rand_sign=torch.randint(0, 2, (K, 4), device=device, generator=generator).float(),
            rand_part=torch.rand(K, 4, device=device, generator=generator))

    def __call__(self, gt_bboxes, gt_labels, label_enc, img_metas, rnd=None):
        assert gt_labels is not None and label_enc is not None and img_metas is not None
        assert len(gt_bboxes) == len(gt_labels)
        B = len(gt_bboxes)
        device = gt_bboxes[0].device
        boxes_n = []
        for meta, b in zip(img_metas, gt_bboxes):
            ih, iw = meta['img_shape'][:2]
            boxes_n.append(ops.bbox_xyxy_to_cxcywh(b) / b.new_tensor([iw, ih, iw, ih]))
        known_num = [int(l.shape[0]) for l in gt_labels]  # host-side sizes: no sync
        max_gt = max(known_num)
        ng = self.get_num_groups(max_gt)
        labels = torch.cat(gt_labels)
        boxes = torch.cat(boxes_n)
        nb = int(boxes.shape[0])
        if rnd is None:
            rnd = self.draw(nb, ng, device)
        batch_idx = torch.cat([torch.full((n,), i, dtype=torch.long, device=device) for i, n in enumerate(known_num)])
        known_labels = labels.repeat(2 * ng)
        known_bid = batch_idx.repeat(2 * ng)
        known_bboxs = boxes.repeat(2 * ng, 1)
        kl, kb = known_labels, known_bboxs
        if self.label_noise_scale > 0:
            kl = torch.where(rnd['label_p'] < self.label_noise_scale * 0.5, rnd['new_label'], known_labels)
        single_pad = max_gt
        pad_size = int(single_pad * 2 * ng)
        if self.box_noise_scale > 0:
            half = known_bboxs[:, 2:] / 2
            xyxy = torch.cat([known_bboxs[:, :2] - half, known_bboxs[:, :2] + half], -1)
            diff = torch.cat([half, half], -1)
            # rows [g*2*nb + nb, (g+1)*2*nb) are the negative copies: rand_part += 1
            neg = ((torch.arange(2 * ng * nb, device=device) // max(nb, 1)) % 2 == 1).to(kb.dtype).unsqueeze(-1)
            part = (rnd['rand_part'] + neg) * (rnd['rand_sign'] * 2.0 - 1.0)
            xyxy = (xyxy + part * diff * self.box_noise_scale).clamp(min=0.0, max=1.0)
            kb = torch.cat([(xyxy[:, :2] + xyxy[:, 2:]) / 2, xyxy[:, 2:] - xyxy[:, :2]], -1)
        input_label_embed = label_enc(kl.long())
        input_bbox_embed = inverse_sigmoid(kb, eps=1e-3)
        q_label = torch.zeros(B, pad_size, self.hidden_dim, device=device)
        q_bbox = torch.zeros(B, pad_size, 4, device=device)
        if nb:
            mki = np.concatenate([np.arange(n) for n in known_num])
            mki = np.concatenate([mki + single_pad * i for i in range(2 * ng)])
            mki = torch.from_numpy(mki).to(device)
            q_label = q_label.index_put((known_bid, mki), input_label_embed)
            q_bbox = q_bbox.index_put((known_bid, mki), input_bbox_embed)
        tgt = pad_size + self.num_queries
        am = np.zeros((tgt, tgt), dtype=bool)
        am[pad_size:, :pad_size] = True
        for i in range(ng):
            lo, hi = single_pad * 2 * i, single_pad * 2 * (i + 1)
            am[lo:hi, hi:pad_size] = True
            am[lo:hi, :lo] = True
        attn_mask = torch.from_numpy(am).to(device)
        return q_label, q_bbox, attn_mask, dict(pad_size=pad_size, num_dn_group=ng)

    def static_queries(self, st, label_enc, rnd=None):
        """The denoising queries in slot layout (B, padcap, .): same arithmetic as __call__ per slot.
        `rnd` in the reference's flat order (query_denoising.py:116-144) is gathered into slots; without it
        the draws are made directly in slot layout (device RNG, capturable)."""
        t = st.t
        B, PC = t['slot_src'].shape
        dev = t['slot_src'].device
        # ten random numbers per slot, u = [label_p, new_label, sign x 4, part x 4]: ONE draw of raw uniforms on the device
        # (capturable), or the reference-order draws gathered into slots; label flip, box jitter, inverse sigmoid and the
        # label-embedding lookup then run as one launch (ops.cdn_queries) instead of ~35 element-wise ones
        if rnd is None:
            u, uniform = torch.rand((B, PC, 10), device=dev), True
        else:
            k = t['slot_k']
            u = torch.cat([rnd['label_p'][k].unsqueeze(-1), rnd['new_label'][k].unsqueeze(-1).float(),
                           rnd['rand_sign'][k].float(), rnd['rand_part'][k]], -1)
            uniform = False
        q_label, q_bbox = ops.cdn_queries(label_enc.weight, t['gt_lab'].reshape(-1), t['gt_boxn'].reshape(-1, 4),
                                          t['slot_src'], t['slot_valid'], t['slot_neg'], u, uniform,
                                          self.label_noise_scale, self.box_noise_scale, self.num_classes)
        return q_label, q_bbox, t['attn_mask'], dict(pad_size=PC, num_dn_group=st.ng)


# ------------------------------------------------------------------------------------------
# shape-static form of a det batch (ground truth + CDN layout): what makes the det iteration
# capturable in one hipGraph and free of host round trips
# ------------------------------------------------------------------------------------------
def _round_up(x, m):
    return (max(int(x), 1) + m - 1) // m * m


_DEBUG_GT = os.environ.get('RSCOTR_DEBUG_GT') == '1'


class DetStatic:
    """Everything of a det batch whose SHAPE follows the ground-truth count in the reference
    (query_denoising.py:55-201 pads to 2*groups*max_gt queries; detr_head.py:475-543 matches (Q, G_i)
    problems image by image), repacked into tensors whose shapes depend only on two capacities:
    `gcap` (ground truths per image, a multiple of 32) and `padcap` (denoising slots per image).

    Denoising slot j of image b is copy i = j // max_gt (even = positive, odd = negative copy; group
    i // 2) of ground truth t = j % max_gt — exactly the reference's `map_known_indice`.  Slots with
    t >= G_b are the reference's own zero padding; slots j >= pad_size = 2*groups*max_gt are EXTRA: the
    attention mask hides them from every other query (they see only themselves) and they carry zero
    loss weight, so the real queries compute what they compute in the reference.

    All tensors are built on the host from the ground-truth COUNTS (known without a device sync) and
    uploaded; `update_into` refreshes a captured iteration's static copies."""

    KEYS = ('gt_box', 'gt_lab', 'gt_boxn', 'gcount', 'gcount_s', 'factors', 'slot_src', 'slot_valid', 'slot_neg', 'slot_inpad',
            'slot_pos', 'slot_k', 'attn_mask', 'norms', 'norms_r', 'scales', 'dn_lab', 'dn_bt', 'dn_bw', 'dn_cw')

    def __init__(self, head, gt_bboxes, gt_labels, img_metas, device, gcap=None, padcap=None, pinned=None, gt_host=None,
                 norms_r_host=None):
        gen = head.dn_generator
        B = len(gt_bboxes)
        Q = head.num_query
        counts = [int(l.shape[0]) for l in gt_labels]
        max_gt = max(counts) if counts else 0
        self.counts, self.max_gt = counts, max_gt
        self.gcap = gcap or _round_up(max_gt, 32)
        ng = gen.get_num_groups(max_gt)
        self.ng, self.single_pad = ng, max_gt
        self.pad_size = int(max_gt * 2 * ng)
        if padcap is None:  # dynamic groups: 2*groups*max_gt <= 2*num_dn_queries whenever max_gt <= num_dn_queries
            padcap = _round_up(self.pad_size, 8)
            if gen.dynamic_dn_groups:
                padcap = max(2 * gen.num_dn, padcap)
        self.padcap = padcap
        assert max_gt <= self.gcap and self.pad_size <= self.padcap
        assert max_gt <= Q, 'the static det path needs num_gt <= num_query'
        G, PC = self.gcap, self.padcap
        shapes = [tuple(m['img_shape'][:2]) for m in img_metas]
        self.img_shapes = shapes
        factors_np = np.asarray([[w, h, w, h] for (h, w) in shapes], dtype=np.float32)
        # host-built layout
        goff = np.concatenate([[0], np.cumsum(counts)])[:-1]
        nb = int(sum(counts))
        j = np.arange(PC)
        sp = max(max_gt, 1)
        copy, t = j // sp, j % sp
        in_pad = j < self.pad_size
        slot_src = np.zeros((B, PC), dtype=np.int64)
        slot_valid = np.zeros((B, PC), dtype=np.float32)
        slot_k = np.zeros((B, PC), dtype=np.int64)
        for b in range(B):
            v = in_pad & (t < counts[b])
            slot_valid[b] = v
            slot_src[b] = np.where(v, b * G + t, b * G)
            slot_k[b] = np.where(v, copy * nb + goff[b] + t, 0)
        slot_neg = np.broadcast_to(((copy % 2) == 1).astype(np.float32), (B, PC)).copy()
        slot_inpad = np.broadcast_to(in_pad.astype(np.float32), (B, PC)).copy()
        slot_pos = slot_valid * (1.0 - slot_neg)
        tgt = PC + Q
        am = np.zeros((tgt, tgt), dtype=bool)
        am[:, :PC] = True                       # nobody sees a denoising slot ...
        for gi in range(ng):                    # ... except the slots of its own group
            lo, hi = sp * 2 * gi, sp * 2 * (gi + 1)
            am[lo:hi, lo:hi] = False
        fake = np.arange(self.pad_size, PC)
        am[fake, :] = True                      # extra slots see only themselves
        am[fake, fake] = False
        norms = self.host_norms(head, counts)
        host = dict(gcount=np.asarray(counts, dtype=np.int32), factors=factors_np, slot_src=slot_src,
                    slot_valid=slot_valid, slot_neg=slot_neg, slot_inpad=slot_inpad, slot_pos=slot_pos, slot_k=slot_k,
                    attn_mask=am, norms=norms)
        # loss_weight / (max(normaliser, 1) + eps) of the three losses, for the matching part (row 0) and the denoising
        # part (row 1) (detr_head.py:379-396): six numbers known with the counts — on one rank they are computed on the
        # host with the fp32 operations torch would run (clamp, add, reciprocal, multiply) and ride the upload
        w3 = (head.loss_cls.loss_weight, head.loss_bbox.loss_weight, head.loss_iou.loss_weight)
        one_rank = ops.dist_world() == 1
        if not one_rank and norms_r_host is None:
            # several ranks: the rank-averaged normalisers come through the HOST control group when the runner made one (the
            # counts are host data: no device collective, no clone, and the batch stays ONE uploaded block as on one rank)
            r = self.reduce_norms_host(norms)
            norms_r_host = None if r is None else r[:4]
        host_scales = one_rank or norms_r_host is not None
        if host_scales:
            f32 = np.float32
            nr = norms if one_rank else np.asarray(norms_r_host, dtype=np.float32)  # (reduce_mean is the identity on one rank)
            cavg = nr if (one_rank or head.sync_cls_avg_factor) else norms
            rows = []
            for ci, pi in ((0, 1), (2, 3)):
                rc = f32(1.0) / (np.maximum(cavg[ci], f32(1.0)) + f32(FP32_EPS))
                rp = f32(1.0) / (np.maximum(nr[pi], f32(1.0)) + f32(FP32_EPS))
                rows.append([rc * f32(w3[0]), rp * f32(w3[1]), rp * f32(w3[2])])
            host['scales'] = np.asarray(rows, dtype=np.float32)
            host['norms_r'] = nr.copy()
        # ground truth, padded with a harmless dummy box.  When the caller hands over host copies of the ground truth
        # (`gt_host` = (boxes, labels) as lists of NumPy arrays: batch['gt_bboxes_host'] / ['gt_labels_host'] of
        # rscotr_amd.synth / rscotr_amd.pipeline, which build them on the host anyway) the padded tensors are laid out on the
        # host too and EVERYTHING of this batch is one pinned block -> one upload (and one copy into a captured iteration's
        # static block) instead of ~45 small launches per det iteration.  The host copies must describe the device tensors:
        # counts and shapes are checked here, the values too under RSCOTR_DEBUG_GT=1 (a device read-back per image)
        hb = hl = [None]
        if gt_host is not None:
            hb, hl = [np.asarray(x) for x in gt_host[0]], [np.asarray(x) for x in gt_host[1]]
            assert len(hb) == len(hl) == B, ('host ground truth of another batch', len(hb), len(hl), B)
            for b in range(B):
                assert hb[b].reshape(-1, 4).shape[0] == counts[b] == hl[b].reshape(-1).shape[0] and \
                    tuple(gt_bboxes[b].shape) == (counts[b], 4), \
                    f'image {b}: host ground truth ({hb[b].shape}, {hl[b].shape}) does not describe the device tensors ' \
                    f'({tuple(gt_bboxes[b].shape)}, {tuple(gt_labels[b].shape)})'
                if _DEBUG_GT:
                    assert np.array_equal(hb[b].reshape(-1, 4).astype(np.float32), gt_bboxes[b].detach().cpu().numpy()) and \
                        np.array_equal(hl[b].reshape(-1), gt_labels[b].detach().cpu().numpy()), f'image {b}: host / device GT differ'
        self.blob = self.host_blob = None
        if all(x is not None for x in hb + hl):
            gt_box = np.zeros((B, G, 4), dtype=np.float32)
            gt_box[:, :, 2:] = 1.0
            gt_lab = np.zeros((B, G), dtype=np.int64)
            for b in range(B):
                if counts[b]:
                    gt_box[b, :counts[b]] = np.asarray(hb[b], dtype=np.float32).reshape(-1, 4)
                    gt_lab[b, :counts[b]] = np.asarray(hl[b], dtype=np.int64).reshape(-1)
            q = gt_box / factors_np[:, None, :]  # bbox_xyxy_to_cxcywh(gt_box / factors) in fp32, as ops does it
            x1, y1, x2, y2 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
            gt_boxn = np.stack([(x1 + x2) / np.float32(2), (y1 + y2) / np.float32(2), x2 - x1, y2 - y1], -1)
            host.update(gt_box=gt_box, gt_lab=gt_lab, gt_boxn=gt_boxn)
            # denoising targets (dino_head.py:323-365: by construction, in slot layout) for the nl decoder layers and the
            # per-set ground-truth counts of the matcher: functions of the ground truth alone, so they ride the same block
            nl = head.transformer.decoder.num_layers
            lab_slot = gt_lab.reshape(-1)[slot_src]
            dn_lab = np.where(slot_pos > 0, lab_slot, head.num_classes).astype(np.int64)
            dn_bt = (gt_boxn.reshape(-1, 4)[slot_src] * slot_pos[..., None]).astype(np.float32)
            dn_bw = np.broadcast_to(slot_pos[..., None], (B, PC, 4))
            rep = lambda a: np.ascontiguousarray(np.broadcast_to(a[None], (nl,) + a.shape))
            host.update(dn_lab=rep(dn_lab), dn_bt=rep(dn_bt), dn_bw=rep(np.ascontiguousarray(dn_bw, dtype=np.float32)),
                        dn_cw=rep(slot_inpad), gcount_s=np.tile(np.asarray(counts, dtype=np.int32), nl + 1))
            self._pack(host, device, pinned)
        else:
            gt_box = torch.zeros((B, G, 4), device=device)
            gt_box[:, :, 2:] = 1.0
            gt_lab = torch.zeros((B, G), dtype=torch.long, device=device)
            for b in range(B):
                if counts[b]:
                    gt_box[b, :counts[b]] = gt_bboxes[b]
                    gt_lab[b, :counts[b]] = gt_labels[b]
            self.t = dict(gt_box=gt_box, gt_lab=gt_lab)
            for k, v in host.items():
                self.t[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device, non_blocking=True)
            f = self.t['factors']
            self.t['gt_boxn'] = ops.bbox_xyxy_to_cxcywh(gt_box / f[:, None, :])
            # the same key set as the host-packed path (a captured iteration refreshes its static tensors key by key:
            # update_into): denoising targets in slot layout (dino_head.py:323-365) and the per-set ground-truth counts
            nl = head.transformer.decoder.num_layers
            t = self.t
            lab_slot = gt_lab.reshape(-1)[t['slot_src']]
            pos = t['slot_pos']
            rep = lambda a: a[None].expand(nl, *a.shape).contiguous()
            t['dn_lab'] = rep(torch.where(pos > 0, lab_slot, torch.full_like(lab_slot, head.num_classes)))
            t['dn_bt'] = rep(t['gt_boxn'].reshape(-1, 4)[t['slot_src']] * pos.unsqueeze(-1))
            t['dn_bw'] = rep(pos.unsqueeze(-1).expand(-1, -1, 4))
            t['dn_cw'] = rep(t['slot_inpad'])
            t['gcount_s'] = t['gcount'].repeat(nl + 1)
        if not host_scales:
            # (no host control group: a bare process group without the runner) every reduce_mean of the reference's det losses (detr_head.py:379-381,389-390; dino_head.py:266-268,
            # 282-283) averages one of these four numbers over the ranks: one small all-reduce, here, outside
            # any captured region
            if 'norms' not in self.t:  # (staging-only block: the rank-local normalisers go up on their own)
                self.t['norms'] = torch.from_numpy(norms).to(device, non_blocking=True)
            self.t['norms_r'] = ops.dist_mean_tensor(self.t['norms']).clone()
            nr, nn_ = self.t['norms_r'], self.t['norms']
            rows = []
            for ci, pi in ((0, 1), (2, 3)):
                cavg = (nr if head.sync_cls_avg_factor else nn_)[ci]
                rc = 1.0 / (cavg.clamp(min=1) + FP32_EPS)
                rp = 1.0 / (nr[pi].clamp(min=1.0) + FP32_EPS)
                rows.append(torch.stack([rc * w3[0], rp * w3[1], rp * w3[2]]))
            self.t['scales'] = torch.stack(rows)

    @staticmethod
    def host_norms(head, counts):
        """The four loss normalisers of a det batch (detr_head.py:379-390, dino_head.py:266-283: cls_avg_factor and num_total_pos
        of the matching part and of the denoising part) — functions of the ground-truth COUNTS alone."""
        Q, B = head.num_query, len(counts)
        max_gt = max(counts) if counts else 0
        ng = head.dn_generator.get_num_groups(max_gt)
        nb = int(sum(counts))
        num_pos = sum(min(Q, g) for g in counts)
        num_neg = B * Q - num_pos
        npos_dn = ng * nb
        bgw = head.bg_cls_weight
        return np.array([num_pos * 1.0 + num_neg * bgw, num_pos, npos_dn * 1.0 + npos_dn * bgw, npos_dn], dtype=np.float32)

    @staticmethod
    def reduce_norms_host(norms, extra=None):
        """reduce_mean of the normalisers over the ranks THROUGH THE HOST control group (one gloo all-reduce; `extra`: more
        floats to SUM in the same message, returned after the four means) — or None where no such group exists."""
        if ops.host_group() is None:
            return None
        import torch.distributed as dist
        w = np.float32(dist.get_world_size())
        vec = np.concatenate([np.asarray(norms, dtype=np.float32) / w, np.asarray(extra if extra is not None else [], dtype=np.float32)])
        return ops.host_allreduce_sum(vec)

    def _pack(self, host, device, pinned=None):
        """All host arrays of the batch in ONE byte block (16-byte aligned fields, KEYS order): `self.t` = typed views of
        the device block.  `pinned`: a pinned staging buffer to fill instead of uploading (GraphedTask.run: the block is
        then copied into the captured iteration's static block by update_into)."""
        fields, off = [], 0
        for k in self.KEYS:
            if k in host:
                a = np.ascontiguousarray(host[k])
                fields.append((k, off, a))
                off += (a.nbytes + 15) // 16 * 16
        self.layout = tuple((k, o, a.dtype.str, a.shape) for k, o, a in fields)
        if pinned is not None and pinned.numel() >= off:
            stage = pinned[:off]
        else:
            stage = torch.empty(max(off, 16), dtype=torch.uint8)
            if device.type == 'cuda':
                stage = stage.pin_memory()
            stage = stage[:off]
        buf = stage.numpy()
        for k, o, a in fields:
            buf[o:o + a.nbytes] = a.view(np.uint8).reshape(-1)
        self.host_blob = stage
        if pinned is not None and pinned.numel() >= off:
            self.t = {}  # (staging only: the tensors of the captured iteration are refreshed by update_into)
            return
        self.blob = stage.to(device, non_blocking=True)
        self.t = {}
        for k, o, a in fields:
            td = torch.bool if a.dtype == np.bool_ else getattr(torch, a.dtype.name)
            self.t[k] = self.blob[o:o + a.nbytes].view(td).view(a.shape)

    def key(self):
        return (self.gcap, self.padcap, tuple(self.img_shapes))

    def update_into(self, static):
        """Copy this batch's tensors into the static tensors of a captured iteration (same capacities): one copy of the
        packed block when both sides are packed the same way."""
        assert static.key() == self.key()
        if self.host_blob is not None and static.blob is not None and self.layout == static.layout:
            static.blob.copy_(self.host_blob, non_blocking=True)
            static.host_keep = self.host_blob  # (the pinned source stays alive until the next refresh)
            packed = {f[0] for f in self.layout}
            for k in self.KEYS:                # tensors outside the block (several ranks: rank-averaged normalisers)
                if k in self.t and k not in packed:
                    static.t[k].copy_(self.t[k], non_blocking=True)
        else:
            missing = [k for k in self.KEYS if (k in static.t) != (k in self.t)]
            if missing:
                raise RuntimeError(f'DetStatic.update_into: the captured iteration and this batch hold different tensors {missing}')
            for k in self.KEYS:
                if k in self.t:
                    static.t[k].copy_(self.t[k], non_blocking=True)
        for a in ('counts', 'max_gt', 'ng', 'single_pad', 'pad_size'):
            setattr(static, a, getattr(self, a))


def build_dn_generator(dn_args):
    if dn_args is None:
        return None
    dn_args = dict(dn_args)
    t = dn_args.pop('type')
    if t != 'CdnQueryGenerator':
        raise NotImplementedError(f'{t} is not supported yet')
    return CdnQueryGenerator(**dn_args)


# ------------------------------------------------------------------------------------------
# transformer — transformer.py:31-273
# ------------------------------------------------------------------------------------------
@MODELS.register_module()
class DinoTransformerDecoder(TransformerLayerSequence):
    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.ref_point_head = build_mlp(self.embed_dims * 2, self.embed_dims, self.embed_dims, 2)
        self.norm = nn.LayerNorm(self.embed_dims)

    @staticmethod
    def gen_sineembed_for_position(pos):
        """pos (B,Q,4) -> (B,Q,512), order (y, x, w, h) (transformer.py:43-76)."""
        scale = 2 * math.pi
        dim_t = torch.arange(128, dtype=torch.float32, device=pos.device)
        dim_t = 10000 ** (2 * (dim_t // 2) / 128)
        outs = []
        for i in (1, 0, 2, 3):
            p = (pos[:, :, i] * scale)[:, :, None] / dim_t
            outs.append(torch.stack((p[:, :, 0::2].sin(), p[:, :, 1::2].cos()), dim=3).flatten(2))
        return torch.cat(outs, dim=2)

    def forward(self, query, value, reference_points, valid_ratios, reg_branches, attn_mask, key_padding_mask, geom,
                unit_ratios=False):
        """Batch-first. Returns (normed outputs of the nl layers, each (B,Q,C); the nl+1 reference sets, each (B,Q,4)) as
        LISTS: the head applies a different branch to every layer's output, and indexing a stacked tensor would cost a
        full-size zero-fill + copy per use and an add per layer in backward (35 launches on 10 MB tensors per step)."""
        output = query
        inter, inter_ref = [], [reference_points]
        values = value if isinstance(value, (list, tuple)) else [value] * len(self.layers)  # (one handle per layer: fan_out)
        vr4 = None if unit_ratios else torch.cat([valid_ratios, valid_ratios], -1)[:, None]
        for lid, layer in enumerate(self.layers):
            assert reference_points.shape[-1] == 4
            if unit_ratios:  # nothing padded: the valid ratios are ones, every level sees the reference points themselves
                rp_in = reference_points[:, :, None]        # (B, Q, 1, 4): shared by the levels (no product, no copy)
                pos_in = ops.sine_embed4(reference_points)
            else:
                rp_in = reference_points[:, :, None] * vr4
                pos_in = ops.sine_embed4(rp_in[:, :, 0, :])
            # query_pos = ref_point_head(...); the layer's first attention adds it to `output`: that sum leaves the MLP's last
            # epilogue as a second output (values only) instead of an element-wise add inside the attention wrapper
            head = [(m.weight, m.bias) for m in self.ref_point_head if isinstance(m, nn.Linear)]
            if ops.STATE.pos_sum:
                query_pos, q_sum = ops.mlp(pos_in, head, act='relu', sum_with=output)
            else:
                query_pos, q_sum = ops.mlp(pos_in, head, act='relu'), None
            output = layer(output, None, values[lid], query_pos=query_pos, attn_masks=attn_mask, query_sum=q_sum,
                           key_padding_mask=key_padding_mask, reference_points=rp_in, **geom.kwargs())
            # three consumers of a layer's output: the next layer, the box branch, the shared norm
            last = lid == len(self.layers) - 1
            output, o_reg, o_norm = ops.fan_out(output, 3) if not last else (output, output, output)
            tmp = _mlp(o_reg, reg_branches[lid])
            new_ref = ops.refine_box(tmp, reference_points, eps=1e-3)
            reference_points = new_ref.detach()
            inter.append(ops.layer_norm(o_norm, self.norm.weight, self.norm.bias))
            inter_ref.append(new_ref)  # look-forward-twice: un-detached
        return inter, inter_ref


@MODELS.register_module()
class DinoTransformer(nn.Module):
    def __init__(self, decoder=None, as_two_stage=False, num_feature_levels=4, two_stage_num_proposals=300,
                 init_cfg=None, **kwargs):
        super().__init__()
        self.decoder = MODELS.build(decoder)
        self.as_two_stage, self.num_feature_levels = as_two_stage, num_feature_levels
        self.two_stage_num_proposals = two_stage_num_proposals
        self.embed_dims = self.decoder.embed_dims
        self.level_embeds = nn.Parameter(torch.Tensor(num_feature_levels, self.embed_dims))
        self.enc_output = nn.Linear(self.embed_dims, self.embed_dims)
        self.enc_output_norm = nn.LayerNorm(self.embed_dims)
        self.query_embed = nn.Embedding(two_stage_num_proposals, self.embed_dims)
        self._geom_cache = {}
        self._pos_cache = {}

    def init_weights(self):
        from .layers import MultiScaleDeformableAttention
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MultiScaleDeformableAttention):
                m.init_weights()
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.query_embed.weight.data)

    @staticmethod
    def get_valid_ratio(mask):
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)

    @staticmethod
    def get_reference_points(shapes, valid_ratios, device):
        ref_list = []
        for lvl, (H, W) in enumerate(shapes):
            ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device),
                                    torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device), indexing='ij')
            ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
            rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
            ref_list.append(torch.stack((rx, ry), -1))
        ref = torch.cat(ref_list, 1)
        return ref[:, :, None] * valid_ratios[:, None]

    @staticmethod
    def gen_proposals(shapes, mask_flat, device):
        """Geometry half of gen_encoder_output_proposals: logit proposals (inf where invalid) and
        the validity mask."""
        B = mask_flat.shape[0]
        props, cur = [], 0
        for lvl, (H, W) in enumerate(shapes):
            mf = mask_flat[:, cur:cur + H * W].view(B, H, W, 1)
            vH = torch.sum(~mf[:, :, 0, 0], 1)
            vW = torch.sum(~mf[:, 0, :, 0], 1)
            gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H, dtype=torch.float32, device=device),
                                    torch.linspace(0, W - 1, W, dtype=torch.float32, device=device), indexing='ij')
            grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
            scale = torch.cat([vW.unsqueeze(-1), vH.unsqueeze(-1)], 1).view(B, 1, 1, 2)
            grid = (grid.unsqueeze(0).expand(B, -1, -1, -1) + 0.5) / scale
            wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
            props.append(torch.cat((grid, wh), -1).view(B, -1, 4))
            cur += H * W
        op = torch.cat(props, 1)
        valid = ((op > 0.01) & (op < 0.99)).all(-1, keepdim=True)
        op = torch.log(op / (1 - op))
        op = op.masked_fill(mask_flat.unsqueeze(-1), float('inf')).masked_fill(~valid, float('inf'))
        return op, valid

    def forward(self, mlvl_feats, mlvl_masks, query_embed, mlvl_pos_embeds, dn_label_query, dn_bbox_query, attn_mask,
                encoder, reg_branches=None, cls_branches=None, record=None, **kwargs):
        assert self.as_two_stage and query_embed is None, 'as_two_stage must be True for DINO'
        device = mlvl_feats[0].device
        feat_f, mask_f, shapes = [], [], []
        for lvl, (feat, mask) in enumerate(zip(mlvl_feats, mlvl_masks)):
            shapes.append(tuple(feat.shape[-2:]))
            feat_f.append(feat.flatten(2).transpose(1, 2))
            mask_f.append(mask.flatten(1))
        feat = torch.cat(feat_f, 1)
        mask_flat = torch.cat(mask_f, 1)
        # no padded image in the batch (known on the host): the padding mask is all-False and every masked_fill
        # with it is the identity — skipped (the reference executes them: detr_head.py / transformer.py:183-241)
        unpadded = kwargs.get('unpadded', False)
        # lvl_pos_embed = pos + level_embeds[lvl] per level, concatenated (transformer.py:196-207): one launch over the
        # token-layout encoding (a constant per level shapes when nothing is padded)
        pkey = (tuple(shapes), str(device)) if unpadded else None
        pos_tok = self._pos_cache.get(pkey) if pkey is not None else None
        if pos_tok is None:
            pos_tok = torch.cat([(p[:1] if unpadded else p).flatten(2).transpose(1, 2) for p in mlvl_pos_embeds], 1).contiguous()
            if pkey is not None:
                self._pos_cache[pkey] = pos_tok
        pos = ops.level_embed_add(None, self.level_embeds, [h * w for h, w in shapes], const=pos_tok, batch=feat.shape[0])
        kpm = None if unpadded else mask_flat
        geom = LevelGeometry.get(shapes, device)
        # without padding the valid ratios are ones and the reference points / proposal geometry depend on the level
        # shapes only: built once per (shapes, batch size) instead of ~100 small launches per iteration
        gkey = (tuple(shapes), feat.shape[0], str(device)) if unpadded else None
        cached = self._geom_cache.get(gkey) if gkey is not None else None
        if cached is None:
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in mlvl_masks], 1)
            reference_points = self.get_reference_points(shapes, valid_ratios, device)
            proposals, valid = self.gen_proposals(shapes, mask_flat, device)
            if gkey is not None:
                self._geom_cache[gkey] = (valid_ratios, reference_points, proposals, valid)
        else:
            valid_ratios, reference_points, proposals, valid = cached
        memory = encoder(feat, None, None, query_pos=pos, query_key_padding_mask=kpm,
                         reference_points=reference_points, **geom.kwargs())
        B = memory.shape[0]
        # the encoder memory feeds the proposal branch and the value projection of every decoder layer: one handle each
        mems = ops.fan_out(memory, 1 + self.decoder.num_layers)
        memory, mem_dec = mems[0], mems[1:]
        om = memory if unpadded else memory.masked_fill(mask_flat.unsqueeze(-1), 0.0)
        om = om.masked_fill(~valid, 0.0)
        om = ops.layer_norm(ops.linear(om, self.enc_output.weight, self.enc_output.bias, range_out=False),
                            self.enc_output_norm.weight, self.enc_output_norm.bias)
        nl = self.decoder.num_layers
        enc_cls = ops.linear(om, cls_branches[nl].weight, cls_branches[nl].bias, range_out=False)
        topk = self.two_stage_num_proposals
        N_tok = enc_cls.shape[1]
        if topk <= min(N_tok, ops.DET_PROPOSALS_MAX_K) and N_tok <= ops.DET_PROPOSALS_MAX_N:
            # row maximum, top-k, proposal add, gathers, sigmoid: one launch (and one for the scattered gradients)
            topk_idx, topk_score, topk_unact, topk_anchor = ops.det_proposals(enc_cls, _mlp(om, reg_branches[nl]), proposals, topk)
        else:
            enc_coord = _mlp(om, reg_branches[nl]) + proposals
            topk_idx = torch.topk(enc_cls.max(-1)[0], topk, dim=1)[1]
            topk_score = torch.gather(enc_cls, 1, topk_idx.unsqueeze(-1).expand(-1, -1, enc_cls.shape[-1]))
            topk_unact = torch.gather(enc_coord, 1, topk_idx.unsqueeze(-1).expand(-1, -1, 4))
            topk_anchor = topk_unact.sigmoid()
            topk_unact = topk_unact.detach()
        if record is not None:
            record['topk_idx'] = topk_idx
        query = ops.batch_param(self.query_embed.weight, B)
        if dn_label_query is not None:
            query = torch.cat([dn_label_query, query], dim=1)
        refp = torch.cat([dn_bbox_query, topk_unact], dim=1) if dn_bbox_query is not None else topk_unact
        refp = refp.sigmoid()
        inter_states, inter_refs = self.decoder(query, mem_dec, refp, valid_ratios, reg_branches, attn_mask,
                                                kpm, geom, unit_ratios=unpadded)
        return inter_states, inter_refs, topk_score, topk_anchor


# ------------------------------------------------------------------------------------------
# head — dino_head.py:16-382 on top of mmdet_detr_head
# ------------------------------------------------------------------------------------------
@MODELS.register_module()
class DINOHead(nn.Module):
    def __init__(self, num_classes, in_channels, num_query=100, num_reg_fcs=2, transformer=None,
                 sync_cls_avg_factor=False, positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, normalize=True),
                 loss_cls=None, loss_bbox=dict(type='L1Loss', loss_weight=5.0), loss_iou=dict(type='GIoULoss', loss_weight=2.0),
                 train_cfg=None, test_cfg=dict(max_per_img=100), with_box_refine=False, as_two_stage=False,
                 num_feature_levels=4, dn_cfg=None, init_cfg=None, **kwargs):
        super().__init__()
        assert as_two_stage and with_box_refine, 'as_two_stage and with_box_refine must be True for DINO'
        transformer = dict(transformer)
        if 'two_stage_num_proposals' in transformer:
            assert transformer['two_stage_num_proposals'] == num_query
        else:
            transformer['two_stage_num_proposals'] = num_query
        transformer['as_two_stage'] = as_two_stage
        self.bg_cls_weight = 0
        self.sync_cls_avg_factor = sync_cls_avg_factor
        self.num_query, self.num_classes, self.in_channels, self.num_reg_fcs = num_query, num_classes, in_channels, num_reg_fcs
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.with_box_refine, self.as_two_stage = with_box_refine, as_two_stage
        if train_cfg:
            self.assigner = MODELS.build(train_cfg['assigner'])
        self.loss_cls, self.loss_bbox, self.loss_iou = MODELS.build(loss_cls), MODELS.build(loss_bbox), MODELS.build(loss_iou)
        self.cls_out_channels = num_classes if self.loss_cls.use_sigmoid else num_classes + 1
        self.positional_encoding = MODELS.build(positional_encoding)
        self._zero_masks = {}
        self.transformer = MODELS.build(transformer)
        self.embed_dims = self.transformer.embed_dims
        assert positional_encoding['num_feats'] * 2 == self.embed_dims
        num_pred = self.transformer.decoder.num_layers + 1
        self.cls_branches = nn.ModuleList([nn.Linear(self.embed_dims, self.cls_out_channels) for _ in range(num_pred)])

        def reg_branch():
            layers = []
            for _ in range(num_reg_fcs):
                layers += [nn.Linear(self.embed_dims, self.embed_dims), nn.ReLU()]
            layers.append(nn.Linear(self.embed_dims, 4))
            return nn.Sequential(*layers)

        self.reg_branches = nn.ModuleList([reg_branch() for _ in range(num_pred)])
        self.label_embedding = nn.Embedding(num_classes, self.embed_dims)
        if dn_cfg is not None:
            dn_cfg = dict(dn_cfg, num_classes=num_classes, num_queries=num_query, hidden_dim=self.embed_dims)
        self.dn_generator = build_dn_generator(dn_cfg)

    def init_weights(self):
        """deformable_detr_head.py:82-94."""
        self.transformer.init_weights()
        bias_init = float(-math.log((1 - 0.01) / 0.01))
        for m in self.cls_branches:
            nn.init.constant_(m.bias, bias_init)
        for m in self.reg_branches:
            nn.init.constant_(m[-1].weight, 0)
            nn.init.constant_(m[-1].bias, 0)
        nn.init.constant_(self.reg_branches[0][-1].bias.data[2:], -2.0)
        for m in self.reg_branches:
            nn.init.constant_(m[-1].bias.data[2:], 0.0)

    # -------------------------------------------------------------------------------------
    # shape-static det path (DetStatic): device-side assignment, no host round trip, capturable.  False
    # selects the reference-shaped dynamic path below (also the fallback when num_gt > num_query).
    static_path = True

    def forward_train_static(self, mlvl_feats, img_metas, st, shared_encoder, rnd=None, record=None):
        dn_label_query, dn_bbox_query, attn_mask, dn_meta = self.dn_generator.static_queries(st, self.label_embedding, rnd)
        outs = self(shared_encoder, mlvl_feats, img_metas, dn_label_query, dn_bbox_query, attn_mask, record=record)
        if record is not None:
            # the dynamic path's view of the outputs (denoising part cut to the reference's pad_size)
            p, ps = st.padcap, st.pad_size
            record['det_outs'] = (torch.cat([outs[0][:, :, :ps], outs[0][:, :, p:]], 2),
                                  torch.cat([outs[1][:, :, :ps], outs[1][:, :, p:]], 2), outs[2], outs[3])
        return self.loss_static(*outs, st, dn_meta, record=record)

    def loss_static(self, all_cls_scores, all_bbox_preds, enc_topk_scores, enc_topk_anchors, st, dn_meta, record=None):
        """`loss` on the shape-static batch: same targets, same sums, same normalisers."""
        t = st.t
        m_cls, m_box, dn_cls, dn_box = self.extract_dn_outputs(all_cls_scores, all_bbox_preds, dn_meta)
        nl, B, Q, C = m_cls.shape
        cls_sets = torch.cat([enc_topk_scores[None], m_cls], 0)
        box_sets = torch.cat([enc_topk_anchors[None], m_box], 0)
        S, G = nl + 1, st.gcap
        a = self.assigner
        cost = ops.match_cost_batched(cls_sets.detach(), box_sets.detach(), t['gt_box'], t['gt_lab'], t['factors'],
                                      a.w_cls, a.w_l1, a.w_iou, a.alpha, a.gamma, a.eps)
        gcount_s = t['gcount_s'] if 'gcount_s' in t else t['gcount'].repeat(S)
        qfg = ops.lsap_device(cost.reshape(S * B, Q, G), gcount_s).view(S, B, G)
        if record is not None:
            host = qfg.cpu().numpy()
            for s_ in range(S):
                for i, g in enumerate(st.counts):
                    if g:
                        q = host[s_, i, :g]
                        order = np.argsort(q, kind='stable')
                        record.setdefault('match', {})[(s_, i)] = (q[order].astype(np.int64), order.astype(np.int64))
        # labels / boxes / weights of the assignment for all S x B problems: one launch (ops.det_targets)
        labels, bbox_t, bbox_w = ops.det_targets(qfg, t['gt_lab'], t['gt_boxn'], Q, self.num_classes)
        # (S, 3) = [cls, bbox, iou] sums of every set times the batch's precomputed weight / normaliser (DetStatic.scales)
        m3 = self._set_losses(cls_sets, box_sets, labels, bbox_t, bbox_w, None, None, st.img_shapes,
                              factors=t['factors'], scales=t['scales'][0])  # rows = interm, d0..d{nl-2}, final
        l_cls, l_box, l_iou = m3[:, 0], m3[:, 1], m3[:, 2]
        d = dict()
        d['interm_loss_cls'], d['interm_loss_bbox'], d['interm_loss_iou'] = l_cls[0], l_box[0], l_iou[0]
        d['loss_cls'], d['loss_bbox'], d['loss_iou'] = l_cls[S - 1], l_box[S - 1], l_iou[S - 1]
        for l in range(nl - 1):
            d[f'd{l}.loss_cls'], d[f'd{l}.loss_bbox'], d[f'd{l}.loss_iou'] = l_cls[l + 1], l_box[l + 1], l_iou[l + 1]
        # denoising part: targets by construction (dino_head.py:323-365) in slot layout
        if 'dn_lab' in t and t['dn_lab'].shape[0] == nl:
            # built on the host with the rest of the batch block (DetStatic): they depend on the ground truth only
            dn_lab, dn_bt, dn_bw, dn_cw = t['dn_lab'], t['dn_bt'], t['dn_bw'], t['dn_cw']
        else:
            lab_slot = t['gt_lab'].reshape(-1)[t['slot_src']]
            pos = t['slot_pos']
            dlabels = torch.where(pos > 0, lab_slot, torch.full_like(lab_slot, self.num_classes))
            dbt = t['gt_boxn'].reshape(-1, 4)[t['slot_src']] * pos.unsqueeze(-1)
            dbw = pos.unsqueeze(-1).expand(-1, -1, 4)
            exp = lambda x: x[None].expand(nl, *x.shape)
            dn_lab, dn_bt, dn_bw, dn_cw = exp(dlabels), exp(dbt), exp(dbw), exp(t['slot_inpad'])
        d3 = self._set_losses(dn_cls, dn_box, dn_lab, dn_bt, dn_bw, None, None, st.img_shapes,
                              factors=t['factors'], cls_weight=dn_cw, scales=t['scales'][1])
        l_cls, l_box, l_iou = d3[:, 0], d3[:, 1], d3[:, 2]  # (nl, 3): rows = d0..d{nl-2}, final
        d['dn_loss_cls'], d['dn_loss_bbox'], d['dn_loss_iou'] = l_cls[nl - 1], l_box[nl - 1], l_iou[nl - 1]
        for l in range(nl - 1):
            d[f'd{l}.dn_loss_cls'], d[f'd{l}.dn_loss_bbox'], d[f'd{l}.dn_loss_iou'] = l_cls[l], l_box[l], l_iou[l]
        # the same 39 scalars as ONE vector in key order, for MTL.pack_losses: summing / stacking 39 0-d views one by one
        # costs ~200 tiny launches per iteration (forward and the select / expand / add chain of their backward)
        perm = getattr(self, '_loss_perm', None)
        if perm is None or perm[0].numel() != S or perm[0].device != m3.device:
            perm = self._loss_perm = (torch.tensor([0, S - 1] + list(range(1, S - 1)), device=m3.device),
                                      torch.tensor([nl - 1] + list(range(nl - 1)), device=m3.device))
        d['__packed__'] = torch.cat([m3.index_select(0, perm[0]).reshape(-1), d3.index_select(0, perm[1]).reshape(-1)])
        return d

    def forward_train(self, mlvl_feats, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None,
                      shared_encoder=None, proposal_cfg=None, rnd=None, record=None, static=None, gt_host=None,
                      norms_r_host=None, **kwargs):
        assert proposal_cfg is None, '"proposal_cfg" must be None'
        assert self.dn_generator is not None, '"dn_cfg" must be set'
        if static is None and self.static_path and max([int(l.shape[0]) for l in gt_labels] + [0]) <= min(self.num_query, 256):
            static = DetStatic(self, gt_bboxes, gt_labels, img_metas, mlvl_feats[0].device, gt_host=gt_host,
                               norms_r_host=norms_r_host)
        if static is not None:
            return self.forward_train_static(mlvl_feats, img_metas, static, shared_encoder, rnd=rnd, record=record)
        dn_label_query, dn_bbox_query, attn_mask, dn_meta = self.dn_generator(
            gt_bboxes, gt_labels, self.label_embedding, img_metas, rnd=rnd)
        outs = self(shared_encoder, mlvl_feats, img_metas, dn_label_query, dn_bbox_query, attn_mask, record=record)
        if record is not None:
            record['det_outs'] = outs
        return self.loss(*outs, gt_bboxes, gt_labels, img_metas, dn_meta, gt_bboxes_ignore=gt_bboxes_ignore,
                         record=record)

    def forward(self, encoder, mlvl_feats, img_metas, dn_label_query=None, dn_bbox_query=None, attn_mask=None,
                record=None):
        B = mlvl_feats[0].size(0)
        device = mlvl_feats[0].device
        ih, iw = img_metas[0]['batch_input_shape']
        padded = any(tuple(m['img_shape'][:2]) != (ih, iw) for m in img_metas)
        mlvl_masks, mlvl_pos = [], []
        if padded:
            img_masks = torch.ones((B, ih, iw), device=device)
            for i, m in enumerate(img_metas):
                h, w = m['img_shape'][:2]
                img_masks[i, :h, :w] = 0
        for feat in mlvl_feats:
            h, w = feat.shape[-2:]
            if padded:
                mask = torch.nn.functional.interpolate(img_masks[None], size=(h, w)).to(torch.bool).squeeze(0)
                mlvl_pos.append(self.positional_encoding(mask))
            else:
                mkey = (B, h, w, str(device))
                mask = self._zero_masks.get(mkey)
                if mask is None:  # (a constant: nothing is padded)
                    mask = self._zero_masks[mkey] = torch.zeros((B, h, w), dtype=torch.bool, device=device)
                mlvl_pos.append(self.positional_encoding.unpadded(B, h, w, device))
            mlvl_masks.append(mask)
        hs, inter_references, topk_score, topk_anchor = self.transformer(
            mlvl_feats, mlvl_masks, None, mlvl_pos, dn_label_query, dn_bbox_query, attn_mask, encoder,
            reg_branches=self.reg_branches, cls_branches=self.cls_branches, record=record, unpadded=not padded)
        if dn_label_query is not None and dn_label_query.size(1) == 0:
            hs = [hs[0] + self.label_embedding.weight[0, 0] * 0.0] + list(hs[1:])  # dino_head.py:124-128
        outputs_classes, outputs_coords = [], []
        for lvl in range(len(hs)):
            outputs_classes.append(ops.linear(hs[lvl], self.cls_branches[lvl].weight, self.cls_branches[lvl].bias, range_out=False))
            outputs_coords.append(ops.refine_box(_mlp(hs[lvl], self.reg_branches[lvl]), inter_references[lvl], eps=1e-3))
        return torch.stack(outputs_classes), torch.stack(outputs_coords), topk_score, topk_anchor

    # ---- inference: dino_head.py:79-82 + mmdet_detr_head/detr_head.py:590-682 ------------------------
    def simple_test(self, feats, img_metas, shared_encoder=None, rescale=False):
        outs = self(shared_encoder, feats, img_metas)
        return self.get_bboxes(*outs, img_metas, rescale=rescale)

    def get_bboxes(self, all_cls_scores, all_bbox_preds, enc_topk_scores, enc_topk_anchors, img_metas, rescale=False):
        """Only the last decoder layer is used."""
        cls_scores, bbox_preds = all_cls_scores[-1], all_bbox_preds[-1]
        return [self._get_bboxes_single(cls_scores[i], bbox_preds[i], m['img_shape'], m['scale_factor'], rescale)
                for i, m in enumerate(img_metas)]

    def _get_bboxes_single(self, cls_score, bbox_pred, img_shape, scale_factor, rescale=False):
        """sigmoid -> top-k over (query, class) -> boxes in pixels, clipped, optionally un-scaled."""
        assert len(cls_score) == len(bbox_pred)
        max_per_img = (self.test_cfg or {}).get('max_per_img', self.num_query)
        scores, indexes = cls_score.sigmoid().view(-1).topk(max_per_img)
        det_labels = indexes % self.num_classes
        bbox_pred = bbox_pred[indexes // self.num_classes]
        det_bboxes = ops.bbox_cxcywh_to_xyxy(bbox_pred)
        det_bboxes[:, 0::2] = (det_bboxes[:, 0::2] * img_shape[1]).clamp(min=0, max=img_shape[1])
        det_bboxes[:, 1::2] = (det_bboxes[:, 1::2] * img_shape[0]).clamp(min=0, max=img_shape[0])
        if rescale:
            sf = scale_factor if hasattr(scale_factor, '__len__') else [scale_factor] * 4
            det_bboxes = det_bboxes / det_bboxes.new_tensor(list(sf))
        return torch.cat((det_bboxes, scores.unsqueeze(1)), -1), det_labels

    # -------------------------------------------------------------------------------------
    @staticmethod
    def extract_dn_outputs(all_cls_scores, all_bbox_preds, dn_meta):
        if dn_meta is not None:
            p = dn_meta['pad_size']
            # one split per tensor: its backward is ONE concatenation (two slices cost a full-size zero-fill + copy each
            # and an add where they meet)
            q = all_cls_scores.shape[2] - p
            dn_cls, m_cls = torch.split(all_cls_scores, [p, q], dim=2)
            dn_box, m_box = torch.split(all_bbox_preds, [p, q], dim=2)
            return m_cls, m_box, dn_cls, dn_box
        return all_cls_scores, all_bbox_preds, None, None

    def _match(self, cls_sets, box_sets, gt_bboxes, gt_labels, img_shapes, record=None):
        """Hungarian targets for S prediction sets at once.  cls_sets (S,B,Q,C), box_sets (S,B,Q,4).
        Returns flat index tensors (device): set/batch/query index of every positive and its gt."""
        S, B, Q, _ = box_sets.shape
        a = self.assigner
        costs, rows, cols = [], [], []
        for i in range(B):
            G = int(gt_bboxes[i].shape[0])
            ih, iw = img_shapes[i]
            if G == 0:
                continue
            c = ops.match_cost(cls_sets[:, i].detach(), box_sets[:, i].detach(), gt_bboxes[i], gt_labels[i], iw, ih,
                               a.w_cls, a.w_l1, a.w_iou, a.alpha, a.gamma, a.eps)  # (S,Q,G)
            costs.append(c.reshape(-1))
            rows += [Q] * S
            cols += [G] * S
        if not costs:
            z = torch.zeros(0, dtype=torch.long, device=box_sets.device)
            return z, z, z, z
        flat = torch.cat(costs)
        r_ind, c_ind = ops.lsap_batch(flat, rows, cols)  # lists of numpy arrays, one per problem
        s_idx, b_idx, q_idx, g_idx = [], [], [], []
        k = 0
        for i in range(B):
            if int(gt_bboxes[i].shape[0]) == 0:
                continue
            for s in range(S):
                r, c = r_ind[k], c_ind[k]
                k += 1
                s_idx.append(np.full(r.shape, s, dtype=np.int64))
                b_idx.append(np.full(r.shape, i, dtype=np.int64))
                q_idx.append(r)
                g_idx.append(c)
                if record is not None:
                    record.setdefault('match', {})[(s, i)] = (r.copy(), c.copy())
        packed = torch.from_numpy(np.stack([np.concatenate(x) for x in (s_idx, b_idx, q_idx, g_idx)]))
        packed = packed.to(box_sets.device, non_blocking=True)
        return packed[0], packed[1], packed[2], packed[3]

    def _set_losses(self, cls_sets, box_sets, labels, bbox_targets, bbox_weights, cls_avg, npos, img_shapes,
                    factors=None, cls_weight=None, scales=None):
        """detr_head.py:372-415 for S sets at once. `cls_avg` / `npos` are the (rank-averaged)
        normalisers before clamping. Returns three (S,) tensors; with `scales` (3,) = weight / (clamped normaliser + eps)
        of [cls, bbox, iou] given instead of the normalisers: ONE (S, 3) tensor [cls, bbox, iou] (one stack + one multiply
        instead of a clamp / add / reciprocal / multiply chain per loss)."""
        S, B, Q, C = cls_sets.shape
        if scales is not None:
            assert Q > 0
            raw_cls = ops.sigmoid_focal_loss_sum(cls_sets.reshape(S, B * Q, C), labels.reshape(S, B * Q),
                                                 self.loss_cls.gamma, self.loss_cls.alpha,
                                                 None if cls_weight is None else cls_weight.reshape(S, B * Q))
            l1, gi = ops.box_loss_sums(box_sets, bbox_targets, bbox_weights, factors.view(B, 4), self.loss_iou.eps)
            return torch.stack([raw_cls, l1, gi], 1) * scales
        cls_avg = ops.clamp_min(cls_avg, 1)
        if Q > 0:
            loss_cls = ops.sigmoid_focal_loss_sum(cls_sets.reshape(S, B * Q, C), labels.reshape(S, B * Q),
                                                  self.loss_cls.gamma, self.loss_cls.alpha,
                                                  None if cls_weight is None else cls_weight.reshape(S, B * Q))
            loss_cls = loss_cls * (self.loss_cls.loss_weight / (cls_avg + FP32_EPS))
        else:
            loss_cls = cls_sets.new_zeros(S)
        npos = ops.clamp_min(npos, 1.0)
        if factors is None:
            factors = cls_sets.new_tensor([[w, h, w, h] for (h, w) in img_shapes])
        l1, gi = ops.box_loss_sums(box_sets, bbox_targets, bbox_weights, factors.view(B, 4), self.loss_iou.eps)
        loss_iou = gi * (self.loss_iou.loss_weight / (npos + FP32_EPS))
        loss_bbox = l1 * (self.loss_bbox.loss_weight / (npos + FP32_EPS))
        return loss_cls, loss_bbox, loss_iou

    def loss(self, all_cls_scores, all_bbox_preds, enc_topk_scores, enc_topk_anchors, gt_bboxes_list, gt_labels_list,
             img_metas, dn_meta=None, gt_bboxes_ignore=None, record=None):
        assert gt_bboxes_ignore is None
        device = all_cls_scores.device
        img_shapes = [tuple(m['img_shape'][:2]) for m in img_metas]
        m_cls, m_box, dn_cls, dn_box = self.extract_dn_outputs(all_cls_scores, all_bbox_preds, dn_meta)
        nl, B, Q, C = m_cls.shape
        # set 0 = encoder proposals (interm), sets 1..nl = decoder layers
        cls_sets = torch.cat([enc_topk_scores[None], m_cls], 0)
        box_sets = torch.cat([enc_topk_anchors[None], m_box], 0)
        S = nl + 1
        s_i, b_i, q_i, g_i = self._match(cls_sets, box_sets, gt_bboxes_list, gt_labels_list, img_shapes, record)
        gcounts = [int(g.shape[0]) for g in gt_labels_list]
        goff = np.concatenate([[0], np.cumsum(gcounts)])[:-1]
        gt_lab_all = torch.cat(gt_labels_list)
        gt_box_n = torch.cat([ops.bbox_xyxy_to_cxcywh(b / b.new_tensor([w, h, w, h]))
                              for b, (h, w) in zip(gt_bboxes_list, img_shapes)])
        goff_t = torch.tensor(goff, dtype=torch.long, device=device)
        labels = torch.full((S, B, Q), self.num_classes, dtype=torch.long, device=device)
        bbox_t = torch.zeros((S, B, Q, 4), device=device)
        bbox_w = torch.zeros((S, B, Q, 4), device=device)
        if s_i.numel():
            gg = goff_t[b_i] + g_i
            labels = labels.index_put((s_i, b_i, q_i), gt_lab_all[gg])
            bbox_t = bbox_t.index_put((s_i, b_i, q_i), gt_box_n[gg])
            bbox_w = bbox_w.index_put((s_i, b_i, q_i), torch.ones((gg.shape[0], 4), device=device))
        num_pos = sum(min(Q, g) for g in gcounts)
        num_neg = B * Q - num_pos
        ng_dn = dn_meta['num_dn_group'] if dn_meta is not None else 0
        npos_dn = ng_dn * sum(gcounts)
        # every reduce_mean of the reference (2 per loss_single x 7, 2 per loss_dn_single x 6)
        # reduces one of these numbers: average them across ranks once
        cavg, cavg_dn = num_pos * 1.0 + num_neg * self.bg_cls_weight, npos_dn * 1.0 + npos_dn * self.bg_cls_weight
        if self.sync_cls_avg_factor:
            cavg, npos_r, cavg_dn, npos_dn_r = ops.dist_mean_vec([cavg, num_pos, cavg_dn, npos_dn], device)
        else:
            npos_r, npos_dn_r = ops.dist_mean_vec([num_pos, npos_dn], device)
        l_cls, l_box, l_iou = self._set_losses(cls_sets, box_sets, labels, bbox_t, bbox_w, cavg, npos_r, img_shapes)
        d = dict()
        d['interm_loss_cls'], d['interm_loss_bbox'], d['interm_loss_iou'] = l_cls[0], l_box[0], l_iou[0]
        d['loss_cls'], d['loss_bbox'], d['loss_iou'] = l_cls[S - 1], l_box[S - 1], l_iou[S - 1]
        for l in range(nl - 1):
            d[f'd{l}.loss_cls'], d[f'd{l}.loss_bbox'], d[f'd{l}.loss_iou'] = l_cls[l + 1], l_box[l + 1], l_iou[l + 1]
        if dn_cls is not None:
            # dino_head.py:323-365: targets by construction
            ng = dn_meta['num_dn_group']
            pad = dn_meta['pad_size']
            single_pad = pad // ng
            bi, qi, gi = [], [], []
            for i, G in enumerate(gcounts):
                if G == 0:
                    continue
                t = np.tile(np.arange(G), (ng, 1))
                qi.append(((np.arange(ng) * single_pad)[:, None] + t).reshape(-1))
                gi.append(t.reshape(-1) + goff[i])
                bi.append(np.full(ng * G, i, dtype=np.int64))
            dlabels = torch.full((B, pad), self.num_classes, dtype=torch.long, device=device)
            dbt = torch.zeros((B, pad, 4), device=device)
            dbw = torch.zeros((B, pad, 4), device=device)
            if bi:
                idx = torch.from_numpy(np.stack([np.concatenate(bi), np.concatenate(qi), np.concatenate(gi)])).to(device)
                dlabels = dlabels.index_put((idx[0], idx[1]), gt_lab_all[idx[2]])
                dbt = dbt.index_put((idx[0], idx[1]), gt_box_n[idx[2]])
                dbw = dbw.index_put((idx[0], idx[1]), torch.ones((idx.shape[1], 4), device=device))
            exp = lambda t: t[None].expand(nl, *t.shape)
            # dino_head.py:340: neg_inds = pos_inds + single_pad // 2, so num_total_neg == num_total_pos
            l_cls, l_box, l_iou = self._set_losses(dn_cls, dn_box, exp(dlabels), exp(dbt), exp(dbw), cavg_dn, npos_dn_r,
                                                   img_shapes)
            d['dn_loss_cls'], d['dn_loss_bbox'], d['dn_loss_iou'] = l_cls[nl - 1], l_box[nl - 1], l_iou[nl - 1]
            for l in range(nl - 1):
                d[f'd{l}.dn_loss_cls'], d[f'd{l}.dn_loss_bbox'], d[f'd{l}.dn_loss_iou'] = l_cls[l], l_box[l], l_iou[l]
        return d

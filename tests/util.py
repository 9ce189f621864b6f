"""Shared helpers for the parity tests."""
import copy
import os

import torch

REF_CFG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'multi')
MAIN_CFG = os.path.join(REF_CFG_DIR, 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')


MLVL_CFG = os.path.join(REF_CFG_DIR, 'MTL_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')


def load_model_cfg(tiny=False, path=None):
    from rscotr_amd import Config
    cfg = Config.fromfile(path or MAIN_CFG)
    m = copy.deepcopy(cfg.model)
    if tiny:  # same architecture, fewer queries so that top-k fits a 64x64 input (85 tokens)
        m['bbox_head']['num_query'] = 30
        m['bbox_head']['dn_cfg']['group_cfg']['num_dn_queries'] = 12
    return cfg, m


def build_model(model_cfg, seed=0, perturb=True):
    from rscotr_amd import MODELS
    torch.manual_seed(seed)
    model = MODELS.build(copy.deepcopy(model_cfg))
    model.init_weights()
    if perturb:
        # the reference init zeroes several layers (MSDA offsets/weights, reg branch heads); give
        # them small random values so that parity tests exercise every gradient path
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if float(p.abs().max()) == 0.0 or 'sampling_offsets.weight' in n or 'attention_weights' in n:
                    p.add_(0.02 * torch.randn(p.shape, generator=g))
    model.train()
    return model


def state_to_oracle(model):
    return {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point)
            for k, v in model.state_dict().items()}


def patch_ops_with_oracle(monkeypatch):
    """CPU-only host-logic tests: route the HIP-backed ops of the product through the oracle so the
    module wiring can be checked without a GPU.  The product itself never does this."""
    from oracle import ops as O
    from rscotr_amd import ops

    import torch.nn.functional as F

    def _set(name, fn):
        # the package re-exports its submodules' names: patch the name wherever it is bound, so that composite ops calling a
        # primitive through their own module (ops.patch_embed -> linear) take the patched one too
        import types
        monkeypatch.setattr(ops, name, fn)
        for sub in vars(ops).values():
            if isinstance(sub, types.ModuleType) and sub.__name__.startswith('rscotr_amd.ops.') and hasattr(sub, name):
                monkeypatch.setattr(sub, name, fn)

    def msda(value, ss, lsi, loc, attn):
        return O.msda_sample(value, ss, lsi, loc, attn)

    def _scaled(y, out_scale):
        return y if out_scale is None else y * out_scale.view(-1, *([1] * (y.dim() - 1)))

    def linear(x, w, b=None, act=None, resid=None, out_scale=None, range_out=True):  # (range_out: a launch detail of the HIP path)
        y = _scaled(F.linear(x, w, b), out_scale)
        return y if resid is None else y + resid

    def mlp(x, layers, act='relu', identity=None, out_scale=None, sum_with=None, range_out=True):
        h = x
        for i, (w, b) in enumerate(layers):
            h = F.linear(h, w, b)
            if i < len(layers) - 1:
                h = F.relu(h) if act == 'relu' else F.gelu(h)
        h = _scaled(h, out_scale)
        if sum_with is not None:
            assert identity is None
            return h, (h + sum_with).detach()
        return h if identity is None else identity + h

    def layer_norm(x, w, b, eps=1e-5):
        return F.layer_norm(x, (x.shape[-1],), w, b, eps)

    def layer_norm_sum(x, w, b, add, eps=1e-5):
        y = F.layer_norm(x, (x.shape[-1],), w, b, eps)
        return y, (y + add).detach()

    def patch_merge_norm(x, hw, w, b, eps=1e-5):
        y, hw2 = ops.patch_merge_gather(x, hw)
        return F.layer_norm(y, (y.shape[-1],), w, b, eps), hw2

    def _check_sum(s, a, b):
        # a producer-formed `a + b` (ops.layer_norm_sum, per-level key sums) carries values only: the stand-ins recompute
        # the sum (autograd needs the real one) and check that what was handed over is that sum
        if s is not None:
            assert not s.requires_grad and torch.allclose(s, (a + b).detach(), rtol=0, atol=1e-5)

    def swin_window_attention(x, hw, qkv_w, qkv_b, bias_table, rel_index, proj_w, proj_b, heads, ws, shift,
                              identity=None, out_scale=None):
        from oracle.model import shift_window_msa
        P = {'a.w_msa.qkv.weight': qkv_w, 'a.w_msa.qkv.bias': qkv_b, 'a.w_msa.proj.weight': proj_w,
             'a.w_msa.proj.bias': proj_b, 'a.w_msa.relative_position_bias_table': bias_table}
        y = _scaled(shift_window_msa(x, hw, P, 'a', heads, ws, shift), out_scale)
        return y if identity is None else identity + y

    def group_norm_tokens(x, groups, w, b, eps=1e-5):
        return F.group_norm(x.transpose(1, 2), groups, w, b, eps).transpose(1, 2)

    def conv3x3s2_tokens(x, hw, w):
        B, L, C = x.shape
        y = F.conv2d(x.transpose(1, 2).reshape(B, C, hw[0], hw[1]), w, None, stride=2, padding=1)
        return y.flatten(2).transpose(1, 2), tuple(y.shape[-2:])

    def upsample_ce(seg_logit, label, ignore_index=255):
        up = F.interpolate(seg_logit, size=label.shape[-2:], mode='bilinear', align_corners=False)
        loss = F.cross_entropy(up, label, reduction='none', ignore_index=ignore_index).mean()
        valid = label != ignore_index
        correct = ((up.argmax(1) == label) & valid).sum().float()
        return loss, (correct * 100.0 / (valid.sum().float() + torch.finfo(torch.float32).eps)).reshape(1)

    def lsap_device(cost, gcount):
        from scipy.optimize import linear_sum_assignment
        P, Q, ld = cost.shape
        out = torch.full((P, ld), -1, dtype=torch.int32)
        for p in range(P):
            g = int(gcount[p])
            if g:
                rs, cs = linear_sum_assignment(cost[p, :, :g].detach().double().numpy())
                out[p, torch.from_numpy(cs)] = torch.from_numpy(rs).int()
        return out

    def mha(x, kx, vx, in_w, in_b, out_w, out_b, heads, attn_mask=None, identity=None, mask_mode=None, q_pos=None, k_pos=None,
            q_sum=None, k_sum=None):
        # (ops.mha: the positional adds happen inside; k_pos None in self-attention means q_pos)
        kx = x if kx is None else kx
        vx = kx if vx is None else vx
        if k_pos is None and kx is x:
            k_pos = q_pos
        _check_sum(q_sum, x, q_pos)
        _check_sum(k_sum, kx, k_pos)
        q_in = x if q_pos is None else x + q_pos
        k_in = kx if k_pos is None else kx + k_pos
        v_in = vx
        B, Lq, C = q_in.shape
        Lk, hd = k_in.shape[1], C // heads
        q = F.linear(q_in, in_w[:C], in_b[:C]).view(B, Lq, heads, hd).transpose(1, 2)
        k = F.linear(k_in, in_w[C:2 * C], in_b[C:2 * C]).view(B, Lk, heads, hd).transpose(1, 2)
        v = F.linear(v_in, in_w[2 * C:], in_b[2 * C:]).view(B, Lk, heads, hd).transpose(1, 2)
        s_ = (q * hd ** -0.5) @ k.transpose(-2, -1)
        if attn_mask is not None:
            if attn_mask.dim() == 2:
                m = attn_mask
            elif attn_mask.shape[0] == B and heads > 1:
                m = attn_mask[:, None]
            else:
                m = attn_mask.view(B, heads, Lq, Lk)
            s_ = s_.masked_fill(m, float('-inf'))
        o = (s_.softmax(-1) @ v).transpose(1, 2).reshape(B, Lq, C)
        y = F.linear(o, out_w, out_b)
        return y if identity is None else identity + y

    def seg_attn_mask(mask_pred, target_size, heads):
        am = F.interpolate(mask_pred, target_size, mode='bilinear', align_corners=False)
        am = (am.flatten(2).sigmoid() < 0.5).detach()
        return am & ~am.all(-1, keepdim=True)

    def msda_prep(off, logit, reference_points, offset_norm, L, P):
        B, Nq, H = logit.shape[:3]
        off = off.view(B, Nq, H, L, P, 2)
        aw = logit.softmax(-1).view(B, Nq, H, L, P)
        r = reference_points
        if r.shape[-1] == 2:
            loc = r[:, :, None, :, None, :] + off / offset_norm[None, None, None, :, None, :]
        else:
            loc = r[:, :, None, :, None, :2] + off / P * r[:, :, None, :, None, 2:] * 0.5
        return loc, aw

    def msda_attention(x, q_pos, value, identity, key_padding_mask, reference_points, spatial_shapes, level_start_index,
                       offset_norm, heads, L, P, w_off, b_off, w_aw, b_aw, w_v, b_v, w_o, b_o, q_sum=None):
        # mmcv MultiScaleDeformableAttention.forward (batch-first), as ops._MSDAAttn computes it
        B, Nq, C = x.shape
        _check_sum(q_sum, x, q_pos)
        q = x if q_pos is None else x + q_pos
        val = x if value is None else value
        v = F.linear(val, w_v, b_v)
        if key_padding_mask is not None:
            v = v.masked_fill(key_padding_mask[..., None], 0.0)
        off = F.linear(q, w_off, b_off).view(B, Nq, heads, L * P * 2)
        logit = F.linear(q, w_aw, b_aw).view(B, Nq, heads, L * P)
        loc, aw = msda_prep(off, logit, reference_points, offset_norm, L, P)
        out = O.msda_sample(v.view(B, -1, heads, C // heads), spatial_shapes, level_start_index, loc, aw)
        y = F.linear(out.reshape(B, Nq, C), w_o, b_o)
        return y if identity is None else identity + y

    _set('msda_attention', msda_attention)

    def level_embed_add(x, weight, sizes, const=None, batch=None, row0=0):
        rows = torch.cat([weight[row0 + l].view(1, -1).expand(int(n), -1) for l, n in enumerate(sizes)], 0)[None]
        out = rows if x is None else x + rows
        if const is not None:
            out = out + const.detach()
        B = x.shape[0] if x is not None else (batch or const.shape[0])
        return out.expand(B, -1, -1) if out.shape[0] != B else out

    _set('level_embed_add', level_embed_add)
    _set('cdn_queries', cdn_queries_ref)

    def match_cost_batched(cls_score, bbox_pred, gt_bboxes, gt_labels, factors, w_cls, w_l1, w_iou, alpha, gamma, eps):
        S, B, Q, C = cls_score.shape
        G = gt_bboxes.shape[1]
        p = cls_score.sigmoid()
        neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
        pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
        idx = gt_labels[None, :, None, :].expand(S, B, Q, G)
        c_cls = torch.gather(pos - neg, 3, idx) * w_cls
        gt_c = ops.bbox_xyxy_to_cxcywh(gt_bboxes / factors[:, None, :])
        c_l1 = (bbox_pred[:, :, :, None, :] - gt_c[None, :, None, :, :]).abs().sum(-1) * w_l1
        boxes = ops.bbox_cxcywh_to_xyxy(bbox_pred) * factors[None, :, None, :]
        c_iou = -ops._giou(boxes, gt_bboxes[None].expand(S, -1, -1, -1), aligned=False) * w_iou
        return c_cls + c_l1 + c_iou

    def sigmoid_focal_loss_sum(pred, target, gamma, alpha, weight=None):
        S, N, C = pred.shape
        p = torch.sigmoid(pred)
        onehot = F.one_hot(target, C + 1)[..., :C].to(pred.dtype)
        tiny = torch.finfo(torch.float32).tiny
        term_p = (1 - p).pow(gamma) * torch.log(p.clamp(min=tiny))
        term_n = p.pow(gamma) * torch.log((1 - p).clamp(min=tiny))
        loss = -onehot * alpha * term_p - (1 - onehot) * (1 - alpha) * term_n
        if weight is not None:
            loss = loss * weight.unsqueeze(-1)
        return loss.sum(dim=(1, 2))

    def box_loss_sums(pred, target, weight, factors, eps=1e-6):
        f = factors.view(1, -1, 1, 4)
        l1 = ((pred - target).abs() * weight).flatten(1).sum(1)
        gi = ((1 - ops._giou(ops.bbox_cxcywh_to_xyxy(pred) * f, ops.bbox_cxcywh_to_xyxy(target) * f, aligned=True, eps=eps))
              * weight.mean(-1)).flatten(1).sum(1)
        return l1, gi

    def mask_logits(e, mask_tokens):
        return torch.einsum('bqd,bpd->bqp', e, mask_tokens)

    _set('mask_logits', mask_logits)

    def refine_box(delta, ref, eps=1e-3):
        from rscotr_amd.layers import inverse_sigmoid
        return (delta + inverse_sigmoid(ref, eps=eps)).sigmoid()

    _set('refine_box', refine_box)
    _set('match_cost_batched', match_cost_batched)
    _set('sigmoid_focal_loss_sum', sigmoid_focal_loss_sum)
    _set('box_loss_sums', box_loss_sums)

    def sine_embed4(pos):
        from rscotr_amd.det_head import DinoTransformerDecoder
        return DinoTransformerDecoder.gen_sineembed_for_position(pos)

    _set('sine_embed4', sine_embed4)
    _set('msda_prep', msda_prep)
    _set('seg_attn_mask', seg_attn_mask)
    _set('mha', mha)
    _set('lsap_device', lsap_device)
    _set('upsample_ce', upsample_ce)
    _set('group_norm_tokens', group_norm_tokens)
    _set('conv3x3s2_tokens', conv3x3s2_tokens)
    _set('swin_window_attention', swin_window_attention)
    _set('msda', msda)
    _set('linear', linear)
    _set('mlp', mlp)
    _set('layer_norm', layer_norm)
    _set('layer_norm_sum', layer_norm_sum)

    def det_proposals(enc_cls, enc_reg, proposals, K):
        enc_coord = enc_reg + proposals
        idx = torch.topk(enc_cls.max(-1)[0], K, dim=1)[1]
        score = torch.gather(enc_cls, 1, idx.unsqueeze(-1).expand(-1, -1, enc_cls.shape[-1]))
        unact = torch.gather(enc_coord, 1, idx.unsqueeze(-1).expand(-1, -1, 4))
        return idx, score, unact.detach(), unact.sigmoid()

    _set('det_proposals', det_proposals)

    def det_targets(q_for_gt, gt_lab, gt_boxn, Q, num_classes):
        S, B, G = q_for_gt.shape
        qfg = q_for_gt.long()
        idx = torch.where(qfg >= 0, qfg, torch.full_like(qfg, Q))
        idx4 = idx.unsqueeze(-1).expand(-1, -1, -1, 4)
        labels = torch.full((S, B, Q + 1), num_classes, dtype=torch.long).scatter_(2, idx, gt_lab[None].expand(S, -1, -1))[:, :, :Q]
        bt = torch.zeros((S, B, Q + 1, 4)).scatter_(2, idx4, gt_boxn[None].expand(S, -1, -1, -1))[:, :, :Q]
        bw = torch.zeros((S, B, Q + 1, 4)).scatter_(2, idx4, torch.ones((S, B, G, 4)))[:, :, :Q]
        return labels, bt, bw

    _set('det_targets', det_targets)
    _set('batch_param', lambda p, B: p[None].expand(B, *p.shape))
    _set('patch_merge_norm', patch_merge_norm)
    _set('layer_norm_fork', lambda x, w, b, eps=1e-5, lazy=False: (layer_norm(x, w, b, eps), x))


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def cdn_queries_ref(weight, gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u, uniform, label_noise_scale, box_noise_scale,
                    num_classes):
    """Plain PyTorch form of ops.cdn_queries: the slot-layout arithmetic of CdnQueryGenerator (query_denoising.py:104-178) as
    rscotr_amd.det_head computed it op by op before the fused kernel."""
    from rscotr_amd.layers import inverse_sigmoid
    lab, boxn = gt_lab[slot_src], gt_boxn[slot_src]
    label_p, new_label, sign, part_r = u[..., 0], u[..., 1], u[..., 2:6], u[..., 6:10]
    if uniform:
        new_label = (new_label * num_classes).long().clamp(max=num_classes - 1)
        sign = (sign >= 0.5).float()
    kl, kb = lab, boxn
    if label_noise_scale > 0:
        kl = torch.where(label_p < label_noise_scale * 0.5, new_label.long(), lab)
    if box_noise_scale > 0:
        half = boxn[..., 2:] / 2
        xyxy = torch.cat([boxn[..., :2] - half, boxn[..., :2] + half], -1)
        diff = torch.cat([half, half], -1)
        part = (part_r + slot_neg.unsqueeze(-1)) * (sign * 2.0 - 1.0)
        xyxy = (xyxy + part * diff * box_noise_scale).clamp(min=0.0, max=1.0)
        kb = torch.cat([(xyxy[..., :2] + xyxy[..., 2:]) / 2, xyxy[..., 2:] - xyxy[..., :2]], -1)
    valid = slot_valid.unsqueeze(-1)
    q_label = torch.nn.functional.embedding(kl.long(), weight) * valid
    q_bbox = torch.where(valid > 0, inverse_sigmoid(kb, eps=1e-3), torch.zeros_like(kb))
    return q_label, q_bbox

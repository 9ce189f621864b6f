"""Oracle model: functional CPU restatement of MTL.forward_train_{cls,det,seg}.

Test infrastructure only (see oracle/__init__.py).  `P` is a dict of fp32 tensors keyed by the
reference's state-dict names; `cfg` is the `model` dict of
configs/multi/MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py (or a shrunken variant).
All stochastic draws (DropPath, Mixup/CutMix, CDN noise) are explicit inputs.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import ops

LN_EPS = 1e-5


def _ln(x, P, name):
    return F.layer_norm(x, (x.shape[-1],), P[name + '.weight'], P[name + '.bias'], LN_EPS)


def _lin(x, P, name):
    return F.linear(x, P[name + '.weight'], P.get(name + '.bias'))


# ------------------------------------------------------------------------------------------
# Swin-T backbone — mmdet 2.25.1 SwinTransformer (cfg ...potsdam.py:9-25), called at
# models/multi/multitask_learner.py:83
# ------------------------------------------------------------------------------------------
def rel_pos_index(ws):
    coords = torch.arange(ws)
    yy, xx = torch.meshgrid(coords, coords, indexing='ij')
    y = yy.reshape(-1)
    x = xx.reshape(-1)
    return (y[:, None] - y[None, :] + ws - 1) * (2 * ws - 1) + (x[:, None] - x[None, :] + ws - 1)


def window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(win, H, W, ws):
    B = int(win.shape[0] / (H * W / ws / ws))
    x = win.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def shift_window_msa(x, hw, P, pre, heads, ws, shift):
    B, L, C = x.shape
    H, W = hw
    q = x.view(B, H, W, C)
    pad_r = (ws - W % ws) % ws
    pad_b = (ws - H % ws) % ws
    q = F.pad(q, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = q.shape[1], q.shape[2]
    if shift > 0:
        q = torch.roll(q, shifts=(-shift, -shift), dims=(1, 2))
        img_mask = torch.zeros((1, Hp, Wp, 1))
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img_mask[:, hs, wsl, :] = cnt
                cnt += 1
        mw = window_partition(img_mask, ws).view(-1, ws * ws)
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        am = am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))
    else:
        am = None
    win = window_partition(q, ws).view(-1, ws * ws, C)
    # WindowMSA.forward
    Bw, N, _ = win.shape
    hd = C // heads
    qkv = _lin(win, P, pre + '.w_msa.qkv').reshape(Bw, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    qq, kk, vv = qkv[0], qkv[1], qkv[2]
    qq = qq * (hd ** -0.5)
    attn = qq @ kk.transpose(-2, -1)
    table = P[pre + '.w_msa.relative_position_bias_table']
    bias = table[rel_pos_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if am is not None:
        nW = am.shape[0]
        attn = attn.view(Bw // nW, nW, heads, N, N) + am.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = attn.softmax(-1)
    o = (attn @ vv).transpose(1, 2).reshape(Bw, N, C)
    o = _lin(o, P, pre + '.w_msa.proj')
    o = window_reverse(o.view(-1, ws, ws, C), Hp, Wp, ws)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    if pad_r > 0 or pad_b > 0:
        o = o[:, :H, :W, :].contiguous()
    return o.reshape(B, H * W, C)


def _droppath(x, keep, rate):
    """mmcv DropPath: x / keep_prob * floor(keep_prob + U); `keep` (B,) holds the 0/1 floors."""
    if keep is None or rate == 0.0:
        return x
    return x.div(1.0 - rate) * keep.view(-1, *([1] * (x.dim() - 1)))


def swin_forward(img, P, bcfg, drop_keep=None, pre='backbone'):
    """Returns the 4 stage outputs (B,C_i,H_i,W_i). drop_keep: (2*sum(depths), B) 0/1 or None."""
    C0 = bcfg['embed_dims']
    depths = bcfg['depths']
    heads = bcfg['num_heads']
    ws = bcfg['window_size']
    total = sum(depths)
    dpr = [x.item() for x in torch.linspace(0, bcfg.get('drop_path_rate', 0.0), total)]
    # PatchEmbed: conv k4 s4 ('corner' adaptive padding) + LN
    k = bcfg.get('patch_size', 4)
    H, W = img.shape[-2:]
    img = F.pad(img, (0, (k - W % k) % k, 0, (k - H % k) % k))
    x = F.conv2d(img, P[pre + '.patch_embed.projection.weight'], P[pre + '.patch_embed.projection.bias'], stride=k)
    hw = (x.shape[2], x.shape[3])
    x = x.flatten(2).transpose(1, 2)
    x = _ln(x, P, pre + '.patch_embed.norm')
    outs = []
    blk = 0
    for s, depth in enumerate(depths):
        for b in range(depth):
            bp = f'{pre}.stages.{s}.blocks.{b}'
            shift = 0 if b % 2 == 0 else ws // 2
            keep_a = None if drop_keep is None else drop_keep[2 * blk]
            keep_f = None if drop_keep is None else drop_keep[2 * blk + 1]
            idt = x
            y = _ln(x, P, bp + '.norm1')
            y = shift_window_msa(y, hw, P, bp + '.attn', heads[s], ws, shift)
            x = idt + _droppath(y, keep_a, dpr[blk])
            idt = x
            y = _ln(x, P, bp + '.norm2')
            y = _lin(F.gelu(_lin(y, P, bp + '.ffn.layers.0.0')), P, bp + '.ffn.layers.1')
            x = idt + _droppath(y, keep_f, dpr[blk])
            blk += 1
        out = x
        out_hw = hw
        if s < len(depths) - 1:
            # PatchMerging: nn.Unfold(2, stride 2) (channel-major), LN(4C), Linear(4C,2C,no bias)
            B, L, C = x.shape
            y = x.view(B, hw[0], hw[1], C).permute(0, 3, 1, 2)
            y = F.pad(y, (0, hw[1] % 2, 0, hw[0] % 2))
            nh, nw = y.shape[2] // 2, y.shape[3] // 2
            y = F.unfold(y, kernel_size=2, stride=2).transpose(1, 2)
            y = _ln(y, P, f'{pre}.stages.{s}.downsample.norm')
            x = F.linear(y, P[f'{pre}.stages.{s}.downsample.reduction.weight'])
            hw = (nh, nw)
        if s in bcfg.get('out_indices', (0, 1, 2, 3)):
            o = _ln(out, P, f'{pre}.norm{s}')
            outs.append(o.view(-1, out_hw[0], out_hw[1], o.shape[-1]).permute(0, 3, 1, 2).contiguous())
    return outs


# ------------------------------------------------------------------------------------------
# ChannelMapper — mmdet 2.25.1 necks/channel_mapper.py (cfg ...potsdam.py:26-33), input
# backbone_feature[-3:] (models/multi/multitask_learner.py:84)
# ------------------------------------------------------------------------------------------
def neck_forward(feats, P, ncfg, pre='neck'):
    G = ncfg['norm_cfg']['num_groups']
    outs = []
    for i, f in enumerate(feats):
        y = F.conv2d(f, P[f'{pre}.convs.{i}.conv.weight'])
        outs.append(F.group_norm(y, G, P[f'{pre}.convs.{i}.gn.weight'], P[f'{pre}.convs.{i}.gn.bias'], 1e-5))
    n_extra = ncfg['num_outs'] - len(feats)
    for i in range(n_extra):
        src = feats[-1] if i == 0 else outs[-1]
        y = F.conv2d(src, P[f'{pre}.extra_convs.{i}.conv.weight'], stride=2, padding=1)
        outs.append(F.group_norm(y, G, P[f'{pre}.extra_convs.{i}.gn.weight'], P[f'{pre}.extra_convs.{i}.gn.bias'], 1e-5))
    return outs


# ------------------------------------------------------------------------------------------
# mmcv 1.6.1 MultiScaleDeformableAttention / FFN / BaseTransformerLayer
# ------------------------------------------------------------------------------------------
def msda_module(query, value, identity, query_pos, key_padding_mask, reference_points,
                spatial_shapes, level_start_index, P, pre, heads=8, levels=4, points=4):
    """seq-first (Nq,B,C) in / out. spatial_shapes: list of (H,W)."""
    if value is None:
        value = query
    if identity is None:
        identity = query
    if query_pos is not None:
        query = query + query_pos
    query = query.permute(1, 0, 2)
    value = value.permute(1, 0, 2)
    B, Nq, C = query.shape
    Nk = value.shape[1]
    v = _lin(value, P, pre + '.value_proj')
    if key_padding_mask is not None:
        v = v.masked_fill(key_padding_mask[..., None], 0.0)
    v = v.view(B, Nk, heads, -1)
    off = _lin(query, P, pre + '.sampling_offsets').view(B, Nq, heads, levels, points, 2)
    aw = _lin(query, P, pre + '.attention_weights').view(B, Nq, heads, levels * points)
    aw = aw.softmax(-1).view(B, Nq, heads, levels, points)
    ss = torch.as_tensor(spatial_shapes, dtype=torch.long)
    if reference_points.shape[-1] == 2:
        norm = torch.stack([ss[..., 1], ss[..., 0]], -1).to(query.dtype)
        loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = reference_points[:, :, None, :, None, :2] + off / points * reference_points[:, :, None, :, None, 2:] * 0.5
    out = ops.msda_sample(v, spatial_shapes, level_start_index, loc, aw)
    out = _lin(out, P, pre + '.output_proj').permute(1, 0, 2)
    return out + identity


def ffn_module(x, P, pre):
    return x + _lin(ops.relu(_lin(x, P, pre + '.layers.0.0')), P, pre + '.layers.1')


def mha_module(query, key, value, identity, query_pos, key_pos, attn_mask, P, pre, heads=8):
    """mmcv MultiheadAttention wrapper, batch_first=False."""
    if key is None:
        key = query
    if value is None:
        value = key
    if identity is None:
        identity = query
    if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
        key_pos = query_pos
    if query_pos is not None:
        query = query + query_pos
    if key_pos is not None:
        key = key + key_pos
    out = ops.mha(query, key, value, P[pre + '.attn.in_proj_weight'], P[pre + '.attn.in_proj_bias'],
                  P[pre + '.attn.out_proj.weight'], P[pre + '.attn.out_proj.bias'], heads, attn_mask)
    return identity + out


def encoder_forward(x, pos, padding_mask, reference_points, spatial_shapes, level_start_index, P,
                    num_layers, pre='shared_encoder'):
    """DetrTransformerEncoder of BaseTransformerLayer('self_attn','norm','ffn','norm'), post-norm."""
    for l in range(num_layers):
        lp = f'{pre}.layers.{l}'
        x = msda_module(x, None, None, pos, padding_mask, reference_points, spatial_shapes,
                        level_start_index, P, lp + '.attentions.0')
        x = _ln(x, P, lp + '.norms.0')
        x = ffn_module(x, P, lp + '.ffns.0')
        x = _ln(x, P, lp + '.norms.1')
    return x


def level_starts(shapes):
    out, s = [], 0
    for h, w in shapes:
        out.append(s)
        s += h * w
    return out


# ------------------------------------------------------------------------------------------
# cls — models/multi/multitask_learner.py:119-127, cls_head/slvl_cls_head.py:14-23 (+ mmcls
# LinearClsHead, LabelSmoothLoss('original'), BatchMixup/BatchCutMix)
# ------------------------------------------------------------------------------------------
def mlvl_memories(neck_feats, P, pre, T, enc_layers, nlev=4, strides=(4, 8, 16, 32)):
    """The part MlvlSegPixelDecoder.forward (seg_head/pixel_decoder.py:80-160) and MlvlClsPixelDecoder.forward
    (cls_head/pixel_decoder.py:41-117) share: neck levels low -> high resolution, + sine position + level
    embedding, through the shared encoder; returns the per-level memories as (B, C, h, w) maps."""
    B = neck_feats[0].shape[0]
    inputs, poss, shapes, refs = [], [], [], []
    for i in range(nlev):
        f = neck_feats[len(neck_feats) - i - 1]
        h, w = f.shape[-2:]
        mask = torch.zeros((B, h, w), dtype=torch.bool)
        pe = ops.sine_positional_encoding(mask, 128, T, True)
        lvl = P[pre + '.level_encoding.weight'][i]
        poss.append((lvl.view(1, -1, 1, 1) + pe).flatten(2).permute(2, 0, 1))
        inputs.append(f.flatten(2).permute(2, 0, 1))
        shapes.append((h, w))
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.get_default_dtype()), torch.arange(w, dtype=torch.get_default_dtype()), indexing='ij')
        stride = strides[len(neck_feats) - i - 1]
        pts = torch.stack([(xs.reshape(-1) + 0.5) * stride, (ys.reshape(-1) + 0.5) * stride], -1)
        refs.append(pts / (torch.tensor([[w, h]], dtype=torch.get_default_dtype()) * stride))
    x = torch.cat(inputs, 0)
    pos = torch.cat(poss, 0)
    ref = torch.cat(refs, 0)[None, :, None].repeat(B, 1, nlev, 1)
    starts = level_starts(shapes)
    pmask = torch.zeros((B, x.shape[0]), dtype=torch.bool)
    mem = encoder_forward(x, pos, pmask, ref, shapes, starts, P, enc_layers)
    mem = mem.permute(1, 2, 0)
    return [m.reshape(B, -1, shapes[i][0], shapes[i][1])
            for i, m in enumerate(torch.split(mem, [h * w for h, w in shapes], dim=-1))]


def mlvl_cls_token(outs, P, scheme):
    """MlvlClsHead.pre_logits_{1..8} (cls_head/mlvl_cls_head.py:76-119)."""
    gap = lambda f: f.mean(dim=(2, 3))
    proj = lambda seq: (seq @ P['cls_head.out_proj.weight'].t() + P['cls_head.out_proj.bias']).squeeze(-1)
    if scheme in (1, 2):
        return gap(outs[scheme - 1])
    if scheme == 3:
        return torch.cat([f.flatten(2) for f in outs], 2).mean(2)
    if scheme == 4:
        return sum(gap(f) for f in outs) / len(outs)
    if scheme in (5, 6):
        return proj(outs[scheme - 5].flatten(2))
    if scheme == 7:
        return proj(torch.cat([f.flatten(2) for f in outs], 2))
    return proj(torch.stack([gap(f) for f in outs], -1))


def cls_features(feats, neck, P, cfg, enc_layers):
    """SlvlClsHead: GAP of the last backbone map; MlvlClsHead: encoder memories -> scheme token."""
    ccfg = cfg['cls_head']
    if ccfg['type'] == 'MlvlClsHead':
        pd = ccfg.get('pixel_decoder', {})
        T = pd.get('positional_encoding', {}).get('temperature', 10000)
        return mlvl_cls_token(mlvl_memories(neck, P, 'cls_head.pixel_decoder', T, enc_layers), P, ccfg.get('scheme', 5))
    return feats[-1].mean(dim=(2, 3))


def apply_cls_augment(img, gt_label, num_classes, aug):
    """aug: dict(kind='identity'|'mixup'|'cutmix', lam, index (B,), bbox=(y1,y2,x1,x2))."""
    onehot = F.one_hot(gt_label, num_classes).to(torch.get_default_dtype())
    if aug is None or aug['kind'] == 'identity':
        return img, onehot
    idx = aug['index']
    lam = aug['lam']
    if aug['kind'] == 'mixup':
        return lam * img + (1 - lam) * img[idx], lam * onehot + (1 - lam) * onehot[idx]
    y1, y2, x1, x2 = aug['bbox']
    img = img.clone()
    img[:, :, y1:y2, x1:x2] = img[idx, :, y1:y2, x1:x2]
    return img, lam * onehot + (1 - lam) * onehot[idx]


def cls_losses(x, soft_label, P, smooth=0.1):
    """x: (B, in_channels) pre-logits."""
    score = _lin(x, P, 'cls_head.fc')
    C = score.shape[1]
    t = soft_label * (1 - smooth) + smooth / C
    loss = (-t * F.log_softmax(score, dim=-1)).sum(-1)
    return OrderedDict(loss=loss.sum() / score.shape[0]), score


# ------------------------------------------------------------------------------------------
# seg — models/multi/seg_head/pixel_decoder.py:80-171, mask2former_head.py:111-205, mmseg 0.28
# BaseDecodeHead.losses
# ------------------------------------------------------------------------------------------
def seg_forward_head(dec_out, mask_feature, target_size, P, heads=8):
    """Mask2FormerHead.forward_head, scheme 2 (models/multi/seg_head/mask2former_head.py:111-137): dec_out (Q,B,C)
    sequence-first, mask_feature (B,C,h,w) -> (mask_pred (B,Q,h,w), attn_mask bool (B*heads,Q,th*tw)).  Pinned to the
    reference's own function: tests/golden/reference_static.npz."""
    d = _ln(dec_out, P, 'seg_head.transformer_decoder.post_norm').transpose(0, 1)
    me = _lin(ops.relu(_lin(ops.relu(_lin(d, P, 'seg_head.mask_embed.0')), P, 'seg_head.mask_embed.2')), P, 'seg_head.mask_embed.4')
    mp = torch.einsum('bqd,bdhw->bqhw', me, mask_feature)
    am = F.interpolate(mp, target_size, mode='bilinear', align_corners=False)
    am = am.flatten(2).unsqueeze(1).repeat(1, heads, 1, 1).flatten(0, 1)
    am = (am.sigmoid() < 0.5).detach()
    return mp, am


def seg_forward(neck_feats, P, cfg, enc_layers, inject_masks=None):
    """`inject_masks`: optional list (one per decoder layer) of the boolean attention masks to USE
    instead of the ones computed here (the computed ones are still returned): the masks are hard
    `sigmoid < 0.5` decisions, and a parity harness compares them bit-wise separately from the
    continuous part of the step (tests/parity.py)."""
    scfg = cfg['seg_head']
    B = neck_feats[0].shape[0]
    nlev = 4
    outs = mlvl_memories(neck_feats, P, 'seg_head.pixel_decoder',
                         scfg['pixel_decoder']['positional_encoding'].get('temperature', 10000), enc_layers)
    mask_feature = F.conv2d(outs[-1], P['seg_head.pixel_decoder.mask_feature.weight'], P['seg_head.pixel_decoder.mask_feature.bias'])
    # Mask2FormerHead.forward
    Tdec = scfg['positional_encoding'].get('temperature', 10000)
    heads = 8
    dec_in, dec_pos = [], []
    for i in range(nlev):
        m = outs[i]
        d = m.flatten(2).permute(2, 0, 1) + P['seg_head.level_embed.weight'][i].view(1, 1, -1)
        mask = torch.zeros((B,) + m.shape[-2:], dtype=torch.bool)
        dec_in.append(d)
        dec_pos.append(ops.sine_positional_encoding(mask, 128, Tdec, True).flatten(2).permute(2, 0, 1))
    qf = P['seg_head.query_feat.weight'].unsqueeze(1).repeat(1, B, 1)
    qe = P['seg_head.query_embed.weight'].unsqueeze(1).repeat(1, B, 1)

    def forward_head(dec_out, target_size):
        return seg_forward_head(dec_out, mask_feature, target_size, P, heads)

    nl = scfg['transformer_decoder']['num_layers']
    mp, am = forward_head(qf, outs[0].shape[-2:])
    masks = []
    for i in range(nl):
        li = i % nlev
        am = am.clone()
        am[torch.where(am.sum(-1) == am.shape[-1])] = False
        masks.append(am)  # the mask this implementation computes for layer i (after the all-True reset)
        if inject_masks is not None:
            am = inject_masks[i]
        lp = f'seg_head.transformer_decoder.layers.{i}'
        qf = mha_module(qf, dec_in[li], dec_in[li], None, qe, dec_pos[li], am, P, lp + '.attentions.0')
        qf = _ln(qf, P, lp + '.norms.0')
        qf = mha_module(qf, qf, qf, None, qe, qe, None, P, lp + '.attentions.1')
        qf = _ln(qf, P, lp + '.norms.1')
        qf = ffn_module(qf, P, lp + '.ffns.0')
        qf = _ln(qf, P, lp + '.norms.2')
        mp, am = forward_head(qf, outs[(i + 1) % nlev].shape[-2:])
    return mp, masks


def seg_losses(seg_logit, gt_semantic_seg, ignore_index=255):
    lab = gt_semantic_seg.squeeze(1)
    up = F.interpolate(seg_logit, size=lab.shape[-2:], mode='bilinear', align_corners=False)
    ce = F.cross_entropy(up, lab, reduction='none', ignore_index=ignore_index)
    losses = OrderedDict()
    losses['loss_ce'] = ce.mean()
    pred = up.argmax(1)
    valid = lab != ignore_index
    correct = ((pred == lab) & valid).to(torch.get_default_dtype()).sum()
    losses['acc_seg'] = (correct * (100.0 / (valid.sum().item() + ops.FP32_EPS))).reshape(1)
    return losses


# ------------------------------------------------------------------------------------------
# det — models/multi/bbox_head/{query_denoising,dino_head,transformer}.py and
# mmdet_detr_head/detr_head.py
# ------------------------------------------------------------------------------------------
def cdn_num_groups(max_gt, num_dn=100):
    g = 1 if max_gt == 0 else num_dn // max_gt
    return max(int(g), 1)


def cdn_queries(gt_bboxes, gt_labels, img_shapes, P, rnd, num_classes=20, num_queries=600,
                hidden=256, num_dn=100, label_noise=0.5, box_noise=1.0):
    """query_denoising.py:55-201. rnd: dict(label_p (K,), new_label (K,) int64, rand_sign (K,4) in
    {0,1}, rand_part (K,4) in [0,1)) with K = 2*num_groups*sum(G_i)."""
    B = len(gt_bboxes)
    boxes_n = []
    for (ih, iw), b in zip(img_shapes, gt_bboxes):
        factor = b.new_tensor([iw, ih, iw, ih]).unsqueeze(0)
        boxes_n.append(ops.bbox_xyxy_to_cxcywh(b) / factor)
    known_num = [int(l.numel()) for l in gt_labels]
    max_gt = max(known_num)
    ng = cdn_num_groups(max_gt, num_dn)
    labels = torch.cat(gt_labels)
    boxes = torch.cat(boxes_n)
    batch_idx = torch.cat([torch.full_like(t.long(), i) for i, t in enumerate(gt_labels)])
    nb = len(boxes)
    known_labels = labels.repeat(2 * ng, 1).view(-1)
    known_bid = batch_idx.repeat(2 * ng, 1).view(-1)
    known_bboxs = boxes.repeat(2 * ng, 1)
    kl = known_labels.clone()
    kb = known_bboxs.clone()
    if label_noise > 0:
        chosen = rnd['label_p'] < (label_noise * 0.5)
        kl = torch.where(chosen, rnd['new_label'], kl)
    single_pad = max_gt
    pad_size = int(single_pad * 2 * ng)
    positive_idx = torch.arange(nb).unsqueeze(0).repeat(ng, 1) + (torch.arange(ng) * nb * 2).unsqueeze(1)
    positive_idx = positive_idx.flatten()
    negative_idx = positive_idx + nb
    if box_noise > 0:
        xyxy = torch.zeros_like(known_bboxs)
        xyxy[:, :2] = known_bboxs[:, :2] - known_bboxs[:, 2:] / 2
        xyxy[:, 2:] = known_bboxs[:, :2] + known_bboxs[:, 2:] / 2
        diff = torch.zeros_like(known_bboxs)
        diff[:, :2] = known_bboxs[:, 2:] / 2
        diff[:, 2:] = known_bboxs[:, 2:] / 2
        sign = rnd['rand_sign'] * 2.0 - 1.0
        part = rnd['rand_part'].clone()
        part[negative_idx] += 1.0
        part = part * sign
        xyxy = xyxy + part * diff * box_noise
        xyxy = xyxy.clamp(min=0.0, max=1.0)
        kb = torch.cat([(xyxy[:, :2] + xyxy[:, 2:]) / 2, xyxy[:, 2:] - xyxy[:, :2]], -1)
    lab_embed = P['bbox_head.label_embedding.weight'][kl.long()]
    box_embed = ops.inverse_sigmoid(kb, eps=1e-3)
    q_label = torch.zeros(B, pad_size, hidden)
    q_bbox = torch.zeros(B, pad_size, 4)
    if nb:
        mki = torch.cat([torch.arange(n) for n in known_num])
        mki = torch.cat([mki + single_pad * i for i in range(2 * ng)]).long()
        q_label[(known_bid.long(), mki)] = lab_embed
        q_bbox[(known_bid.long(), mki)] = box_embed
    tgt = pad_size + num_queries
    am = torch.zeros(tgt, tgt, dtype=torch.bool)
    am[pad_size:, :pad_size] = True
    for i in range(ng):
        lo, hi = single_pad * 2 * i, single_pad * 2 * (i + 1)
        am[lo:hi, hi:pad_size] = True
        am[lo:hi, :lo] = True
    return q_label, q_bbox, am, dict(pad_size=pad_size, num_dn_group=ng)


def gen_sineembed(pos):
    """transformer.py:43-76 (4-d case)."""
    scale = 2 * math.pi
    dim_t = torch.arange(128, dtype=torch.get_default_dtype())
    dim_t = 10000 ** (2 * (dim_t // 2) / 128)
    outs = []
    for i in (1, 0, 2, 3):  # y, x, w, h
        e = pos[:, :, i] * scale
        p = e[:, :, None] / dim_t
        outs.append(torch.stack((p[:, :, 0::2].sin(), p[:, :, 1::2].cos()), dim=3).flatten(2))
    return torch.cat(outs, dim=2)


def _reg_branch(x, P, i):
    pre = f'bbox_head.reg_branches.{i}'
    return _lin(ops.relu(_lin(ops.relu(_lin(x, P, pre + '.0')), P, pre + '.2')), P, pre + '.4')


def det_forward(neck_feats, img_shapes, batch_shape, P, cfg, enc_layers, dn=None, inject_topk=None, record=None):
    """dino_head.py:84-150 + transformer.py:164-273. dn = (q_label, q_bbox, attn_mask) or None.
    Returns all_cls (6,B,Q,20), all_box (6,B,Q,4), topk_score, topk_anchor."""
    hcfg = cfg['bbox_head']
    nq = hcfg['num_query']
    B = neck_feats[0].shape[0]
    T = hcfg['positional_encoding'].get('temperature', 10000)
    ih, iw = batch_shape
    img_masks = torch.ones((B, ih, iw))
    for i, (h, w) in enumerate(img_shapes):
        img_masks[i, :h, :w] = 0
    masks, poss = [], []
    for f in neck_feats:
        m = F.interpolate(img_masks[None], size=f.shape[-2:]).to(torch.bool).squeeze(0)
        masks.append(m)
        poss.append(ops.sine_positional_encoding(m, 128, T, True))
    feat_f, mask_f, pos_f, shapes = [], [], [], []
    for lvl, (f, m, pe) in enumerate(zip(neck_feats, masks, poss)):
        shapes.append(tuple(f.shape[-2:]))
        feat_f.append(f.flatten(2).transpose(1, 2))
        mask_f.append(m.flatten(1))
        pos_f.append(pe.flatten(2).transpose(1, 2) + P['bbox_head.transformer.level_embeds'][lvl].view(1, 1, -1))
    feat = torch.cat(feat_f, 1)
    mask_flat = torch.cat(mask_f, 1)
    pos = torch.cat(pos_f, 1)
    starts = level_starts(shapes)
    vr = []
    for m in masks:
        _, H, W = m.shape
        vh = torch.sum(~m[:, :, 0], 1).to(torch.get_default_dtype()) / H
        vw = torch.sum(~m[:, 0, :], 1).to(torch.get_default_dtype()) / W
        vr.append(torch.stack([vw, vh], -1))
    valid_ratios = torch.stack(vr, 1)  # (B,L,2)
    ref_list = []
    for lvl, (H, W) in enumerate(shapes):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing='ij')
        ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
        rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
        ref_list.append(torch.stack((rx, ry), -1))
    ref = torch.cat(ref_list, 1)
    ref = ref[:, :, None] * valid_ratios[:, None]
    memory = encoder_forward(feat.permute(1, 0, 2), pos.permute(1, 0, 2), mask_flat, ref, shapes, starts, P, enc_layers)
    memory = memory.permute(1, 0, 2)  # (B,N,C)
    # gen_encoder_output_proposals
    props = []
    cur = 0
    for lvl, (H, W) in enumerate(shapes):
        mf = mask_flat[:, cur:cur + H * W].view(B, H, W, 1)
        vH = torch.sum(~mf[:, :, 0, 0], 1)
        vW = torch.sum(~mf[:, 0, :, 0], 1)
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing='ij')
        grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
        scale = torch.cat([vW.unsqueeze(-1), vH.unsqueeze(-1)], 1).view(B, 1, 1, 2)
        grid = (grid.unsqueeze(0).expand(B, -1, -1, -1) + 0.5) / scale
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        props.append(torch.cat((grid, wh), -1).view(B, -1, 4))
        cur += H * W
    op = torch.cat(props, 1)
    op_valid = ((op > 0.01) & (op < 0.99)).all(-1, keepdim=True)
    op = torch.log(op / (1 - op))
    op = op.masked_fill(mask_flat.unsqueeze(-1), float('inf')).masked_fill(~op_valid, float('inf'))
    om = memory.masked_fill(mask_flat.unsqueeze(-1), 0.0).masked_fill(~op_valid, 0.0)
    om = _ln(_lin(om, P, 'bbox_head.transformer.enc_output'), P, 'bbox_head.transformer.enc_output_norm')
    ndec = hcfg['transformer']['decoder']['num_layers']
    enc_cls = _lin(om, P, f'bbox_head.cls_branches.{ndec}')
    enc_coord = _reg_branch(om, P, ndec) + op
    topk_idx = torch.topk(enc_cls.max(-1)[0], nq, dim=1)[1]
    if record is not None:
        record['topk_idx'] = topk_idx          # the oracle's own decision
        record['topk_scores'] = enc_cls.max(-1)[0].detach()
    if inject_topk is not None:
        # parity harness: the proposal selection is a hard decision (a score within fp32 rounding of the
        # 600th may land on either side); it is compared on its own and the continuous part of the step
        # is then evaluated under the product's selection
        topk_idx = inject_topk
    topk_score = torch.gather(enc_cls, 1, topk_idx.unsqueeze(-1).repeat(1, 1, enc_cls.shape[-1]))
    topk_unact = torch.gather(enc_coord, 1, topk_idx.unsqueeze(-1).repeat(1, 1, 4))
    topk_anchor = topk_unact.sigmoid()
    topk_unact = topk_unact.detach()
    query = P['bbox_head.transformer.query_embed.weight'][:, None, :].repeat(1, B, 1).transpose(0, 1)
    attn_mask = None
    if dn is not None:
        query = torch.cat([dn[0], query], dim=1)
        refp = torch.cat([dn[1], topk_unact], dim=1)
        attn_mask = dn[2]
    else:
        refp = topk_unact
    refp = refp.sigmoid()
    # DinoTransformerDecoder.forward
    out = query.permute(1, 0, 2)
    mem_sf = memory.permute(1, 0, 2)
    inter, inter_ref = [], [refp]
    for lid in range(ndec):
        rp_in = refp[:, :, None] * torch.cat([valid_ratios, valid_ratios], -1)[:, None]
        qse = gen_sineembed(rp_in[:, :, 0, :])
        qpos = _lin(ops.relu(_lin(qse, P, 'bbox_head.transformer.decoder.ref_point_head.0')), P,
                    'bbox_head.transformer.decoder.ref_point_head.2').permute(1, 0, 2)
        lp = f'bbox_head.transformer.decoder.layers.{lid}'
        out = mha_module(out, out, out, None, qpos, qpos, attn_mask, P, lp + '.attentions.0')
        out = _ln(out, P, lp + '.norms.0')
        out = msda_module(out, mem_sf, None, qpos, mask_flat, rp_in, shapes, starts, P, lp + '.attentions.1')
        out = _ln(out, P, lp + '.norms.1')
        out = ffn_module(out, P, lp + '.ffns.0')
        out = _ln(out, P, lp + '.norms.2')
        ob = out.permute(1, 0, 2)
        new_ref = (_reg_branch(ob, P, lid) + ops.inverse_sigmoid(refp, eps=1e-3)).sigmoid()
        refp = new_ref.detach()
        inter.append(_ln(out, P, 'bbox_head.transformer.decoder.norm'))
        inter_ref.append(new_ref)
    hs = torch.stack(inter).permute(0, 2, 1, 3)
    if dn is not None and dn[0].size(1) == 0:
        hs = hs.clone()
        hs[0] += P['bbox_head.label_embedding.weight'][0, 0] * 0.0
    all_cls, all_box = [], []
    for lvl in range(ndec):
        r = ops.inverse_sigmoid(inter_ref[lvl], eps=1e-3)
        all_cls.append(_lin(hs[lvl], P, f'bbox_head.cls_branches.{lvl}'))
        all_box.append((_reg_branch(hs[lvl], P, lvl) + r).sigmoid())
    return torch.stack(all_cls), torch.stack(all_box), topk_score, topk_anchor


def _loss_from_targets(cls_scores, bbox_preds, labels, bbox_targets, bbox_weights, num_pos, num_neg,
                       img_shapes, num_classes, world=1):
    """detr_head.py:372-415 / dino_head.py:261-309 after target assignment."""
    B, Q = bbox_preds.shape[:2]
    cs = cls_scores.reshape(-1, cls_scores.shape[-1])
    cls_avg = max(num_pos * 1.0 + num_neg * 0.0, 1)
    if cs.shape[0] > 0:
        loss_cls = ops.sigmoid_focal_loss_sum(cs, labels, 2.0, 0.25) / (cls_avg + ops.FP32_EPS)
    else:
        loss_cls = torch.zeros(1)
    npos = max(float(num_pos), 1.0)
    factors = torch.cat([bbox_preds.new_tensor([w, h, w, h]).unsqueeze(0).repeat(Q, 1) for (h, w) in img_shapes], 0)
    bp = bbox_preds.reshape(-1, 4)
    boxes = ops.bbox_cxcywh_to_xyxy(bp) * factors
    boxes_gt = ops.bbox_cxcywh_to_xyxy(bbox_targets) * factors
    loss_iou = 2.0 * ops.giou_loss_sum(boxes, boxes_gt, bbox_weights) / (npos + ops.FP32_EPS)
    loss_bbox = 5.0 * ops.l1_loss_sum(bp, bbox_targets, bbox_weights) / (npos + ops.FP32_EPS)
    return loss_cls, loss_bbox, loss_iou


def det_loss_single(cls_scores, bbox_preds, gt_bboxes, gt_labels, img_shapes, num_classes=20, record=None, inject=None):
    """detr_head.py:333-416 incl. Hungarian target assignment (:475-543).  `inject` (tests only): per image a dict with
    the assignment (pos_inds, pos_assigned_gt_inds) to use instead of solving — the fp64 evaluation of a step takes the
    fp32 evaluation's hard decisions so that the continuous part is compared under identical decisions."""
    B, Q = bbox_preds.shape[:2]
    labels_l, bt_l, bw_l = [], [], []
    npos = nneg = 0
    for i in range(B):
        ih, iw = img_shapes[i]
        G = gt_bboxes[i].shape[0]
        labels = torch.full((Q,), num_classes, dtype=torch.long)
        bt = torch.zeros(Q, 4)
        bw = torch.zeros(Q, 4)
        if G > 0:
            cost = ops.match_cost(cls_scores[i].detach(), bbox_preds[i].detach(), gt_bboxes[i], gt_labels[i], iw, ih)
            pos, gti = ops.hungarian_assign(cost)
            if inject is not None:
                pos, gti = inject[i]['pos_inds'], inject[i]['pos_assigned_gt_inds']
        else:
            cost = torch.zeros(Q, 0)
            pos = gti = torch.zeros(0, dtype=torch.long)
        if record is not None:
            record.append(dict(cost=cost, pos_inds=pos, pos_assigned_gt_inds=gti))
        labels[pos] = gt_labels[i][gti]
        bw[pos] = 1.0
        factor = bbox_preds.new_tensor([iw, ih, iw, ih]).unsqueeze(0)
        bt[pos] = ops.bbox_xyxy_to_cxcywh(gt_bboxes[i][gti] / factor)
        npos += pos.numel()
        nneg += Q - pos.numel()
        labels_l.append(labels)
        bt_l.append(bt)
        bw_l.append(bw)
    return _loss_from_targets(cls_scores, bbox_preds, torch.cat(labels_l), torch.cat(bt_l), torch.cat(bw_l),
                              npos, nneg, img_shapes, num_classes)


def det_loss_dn_single(dn_cls, dn_box, gt_bboxes, gt_labels, img_shapes, dn_meta, num_classes=20):
    """dino_head.py:247-365."""
    B, Q = dn_box.shape[:2]
    ng = dn_meta['num_dn_group']
    single_pad = dn_meta['pad_size'] // ng
    labels_l, bt_l, bw_l = [], [], []
    npos = nneg = 0
    for i in range(B):
        ih, iw = img_shapes[i]
        G = gt_labels[i].numel()
        if G > 0:
            t = torch.arange(G).unsqueeze(0).repeat(ng, 1)
            gti = t.flatten()
            pos = ((torch.arange(ng) * single_pad).unsqueeze(1) + t).flatten()
        else:
            pos = gti = torch.zeros(0, dtype=torch.long)
        labels = torch.full((Q,), num_classes, dtype=torch.long)
        labels[pos] = gt_labels[i][gti]
        bt = torch.zeros(Q, 4)
        bw = torch.zeros(Q, 4)
        bw[pos] = 1.0
        factor = dn_box.new_tensor([iw, ih, iw, ih]).unsqueeze(0)
        bt[pos] = ops.bbox_xyxy_to_cxcywh(gt_bboxes[i] / factor).repeat([ng, 1])
        npos += pos.numel()
        nneg += pos.numel()
        labels_l.append(labels)
        bt_l.append(bt)
        bw_l.append(bw)
    return _loss_from_targets(dn_cls, dn_box, torch.cat(labels_l), torch.cat(bt_l), torch.cat(bw_l),
                              npos, nneg, img_shapes, num_classes)


def det_losses(all_cls, all_box, topk_score, topk_anchor, gt_bboxes, gt_labels, img_shapes, dn_meta,
               num_classes=20, record=None, inject_match=None):
    """dino_head.py:152-234.  inject_match: {'interm' | 'dec{l}': [per-image assignment dicts]} (see det_loss_single)."""
    inj = inject_match or {}
    pad = dn_meta['pad_size'] if dn_meta is not None else 0
    m_cls, m_box = all_cls[:, :, pad:], all_box[:, :, pad:]
    d = OrderedDict()
    rec = None if record is None else record.setdefault('interm', [])
    d['interm_loss_cls'], d['interm_loss_bbox'], d['interm_loss_iou'] = det_loss_single(
        topk_score, topk_anchor, gt_bboxes, gt_labels, img_shapes, num_classes, rec, inj.get('interm'))
    n = all_cls.shape[0]
    per = []
    for l in range(n):
        rec = None if record is None else record.setdefault(f'dec{l}', [])
        per.append(det_loss_single(m_cls[l], m_box[l], gt_bboxes, gt_labels, img_shapes, num_classes, rec,
                                   inj.get(f'dec{l}')))
    d['loss_cls'], d['loss_bbox'], d['loss_iou'] = per[-1]
    for l in range(n - 1):
        d[f'd{l}.loss_cls'], d[f'd{l}.loss_bbox'], d[f'd{l}.loss_iou'] = per[l]
    if dn_meta is not None:
        per = [det_loss_dn_single(all_cls[l, :, :pad], all_box[l, :, :pad], gt_bboxes, gt_labels, img_shapes,
                                  dn_meta, num_classes) for l in range(n)]
        d['dn_loss_cls'], d['dn_loss_bbox'], d['dn_loss_iou'] = per[-1]
        for l in range(n - 1):
            d[f'd{l}.dn_loss_cls'], d[f'd{l}.dn_loss_bbox'], d[f'd{l}.dn_loss_iou'] = per[l]
    return d


# ------------------------------------------------------------------------------------------
# MTL.forward_train_* + _parse_losses + train_step (models/multi/multitask_learner.py:119-147,
# 229-245, 274-306), single process
# ------------------------------------------------------------------------------------------
def forward_train(P, cfg, batch, rnd=None, record=None):
    """batch: dict(task, img, + task fields). Returns OrderedDict of raw losses (pre _parse_losses)."""
    rnd = rnd or {}
    task = batch['task']
    enc_layers = cfg['shared_encoder']['num_layers']
    img = batch['img']
    if task == 'cls':
        img, soft = apply_cls_augment(img, batch['gt_label'], cfg['cls_head']['num_classes'], rnd.get('cls_aug'))
    feats = swin_forward(img, P, cfg['backbone'], rnd.get('drop_keep'))
    if record is not None:
        record['backbone_feats'] = feats
    neck = neck_forward(feats[-3:], P, cfg['neck'])
    if record is not None:
        record['neck_feats'] = neck
    if task == 'cls':
        losses, score = cls_losses(cls_features(feats, neck, P, cfg, enc_layers), soft, P,
                                   cfg['cls_head']['loss'].get('label_smooth_val', 0.1))
        if record is not None:
            record['cls_score'] = score
        return losses
    if task == 'seg':
        logit, masks = seg_forward(neck, P, cfg, enc_layers,
                                   inject_masks=None if rnd is None else rnd.get('seg_attn_masks'))
        if record is not None:
            record['seg_logit'] = logit
            record['attn_masks'] = masks
        raw = seg_losses(logit, batch['gt_semantic_seg'])
        return OrderedDict(('seg.' + k, v) for k, v in raw.items())
    img_shapes = [tuple(m['img_shape'][:2]) for m in batch['img_metas']]
    hcfg = cfg['bbox_head']
    dn = None
    dn_meta = None
    if rnd.get('cdn') is not None:
        ql, qb, am, dn_meta = cdn_queries(batch['gt_bboxes'], batch['gt_labels'], img_shapes, P, rnd['cdn'],
                                          hcfg['num_classes'], hcfg['num_query'], 256,
                                          hcfg['dn_cfg']['group_cfg']['num_dn_queries'],
                                          hcfg['dn_cfg']['noise_scale']['label'], hcfg['dn_cfg']['noise_scale']['box'])
        dn = (ql, qb, am)
    outs = det_forward(neck, img_shapes, tuple(img.shape[-2:]), P, cfg, enc_layers, dn,
                       inject_topk=rnd.get('det_topk_idx'), record=record)
    if record is not None:
        record['det_outs'] = outs
        record['match'] = {}
    return det_losses(*outs, batch['gt_bboxes'], batch['gt_labels'], img_shapes, dn_meta, hcfg['num_classes'],
                      None if record is None else record['match'], inject_match=rnd.get('det_match'))


def parse_losses(losses):
    """MTL._parse_losses, single process (models/multi/multitask_learner.py:275-304); pinned to the reference's own
    function: tests/golden/reference_static.npz."""
    log_vars = OrderedDict()
    for k, v in losses.items():
        if isinstance(v, torch.Tensor):
            log_vars[k] = v.mean()
        elif isinstance(v, list):
            log_vars[k] = sum(x.mean() for x in v)
        else:
            raise TypeError(f'{k} is not a tensor or list of tensors')
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, OrderedDict((k, float(v.item())) for k, v in log_vars.items())


def train_step(P, cfg, batch, rnd=None, record=None):
    losses = forward_train(P, cfg, batch, rnd, record)
    loss, log_vars = parse_losses(losses)
    task, ds = batch['task'], batch.get('dataset_name')
    w = cfg.get('task_weight', {}).get(task, 1)
    loss = loss * w
    log_vars = OrderedDict((f'{task}.{ds}.{k}', v * w) for k, v in log_vars.items())
    return dict(loss=loss, log_vars=log_vars, num_samples=len(batch['img_metas']))


# ------------------------------------------------------------------------------------------
# inference — MTL.simple_test_{cls,det,seg} (models/multi/multitask_learner.py:149-227),
# DETRHead._get_bboxes_single (mmdet_detr_head/detr_head.py:627-682), mmcls LinearClsHead.simple_test
# ------------------------------------------------------------------------------------------
def simple_test(P, cfg, task, img, img_metas, rescale=None, inject=None, record=None):
    """cls: (B, num_classes) softmax scores; det: list of (det_bboxes (k,5), det_labels (k,)) per image;
    seg: (B, H, W) arg-max maps.  `inject`: optional dict(det_topk_idx=..., seg_attn_masks=...) of hard decisions."""
    inject = inject or {}
    enc_layers = cfg['shared_encoder']['num_layers']
    feats = swin_forward(img, P, cfg['backbone'], None)
    neck = neck_forward(feats[-3:], P, cfg['neck'])
    if task == 'cls':
        return F.softmax(_lin(cls_features(feats, neck, P, cfg, enc_layers), P, 'cls_head.fc'), dim=-1)
    if task == 'det':
        img_shapes = [tuple(m['img_shape'][:2]) for m in img_metas]
        all_cls, all_box, _, _ = det_forward(neck, img_shapes, tuple(img.shape[-2:]), P, cfg, enc_layers, None,
                                             inject_topk=inject.get('det_topk_idx'), record=record)
        ncls = cfg['bbox_head']['num_classes']
        k = cfg['test_cfg']['det'].get('max_per_img', cfg['bbox_head']['num_query'])
        out = []
        for i, m in enumerate(img_metas):
            scores, idx = all_cls[-1, i].sigmoid().view(-1).topk(k)
            labels = idx % ncls
            box = ops.bbox_cxcywh_to_xyxy(all_box[-1, i][idx // ncls])
            h, w = m['img_shape'][:2]
            box[:, 0::2] = (box[:, 0::2] * w).clamp(min=0, max=w)
            box[:, 1::2] = (box[:, 1::2] * h).clamp(min=0, max=h)
            if rescale:
                sf = m['scale_factor']
                box = box / box.new_tensor(list(sf) if hasattr(sf, '__len__') else [sf] * 4)
            out.append((torch.cat([box, scores[:, None]], -1), labels))
        return out
    logit, masks = seg_forward(neck, P, cfg, enc_layers, inject_masks=inject.get('seg_attn_masks'))
    if record is not None:
        record['attn_masks'] = masks
    logit = F.interpolate(logit, size=img.shape[2:], mode='bilinear', align_corners=False)
    if rescale is None or rescale:
        h, w = img_metas[0]['img_shape'][:2]
        logit = F.interpolate(logit[:, :, :h, :w], size=tuple(img_metas[0]['ori_shape'][:2]), mode='bilinear',
                              align_corners=False)
    return F.softmax(logit, dim=1).argmax(dim=1)

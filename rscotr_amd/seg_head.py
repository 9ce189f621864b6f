"""Segmentation decoder (`type='Mask2FormerHead'` + `MlvlSegPixelDecoder`) over the shared encoder.

Mirrors models/multi/seg_head/pixel_decoder.py:14-170 and mask2former_head.py:18-208 (plus
mmseg 0.28 `BaseDecodeHead.losses` reached at mask2former_head.py:204) with the reference's
parameter names.  Tokens are batch-first.
"""
import copy

import torch
import torch.nn as nn

from . import ops
from .layers import LevelGeometry
from .registry import MODELS


@MODELS.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0,
                 loss_name='loss_ce', avg_non_ignore=False):
        super().__init__()
        assert not use_sigmoid and not use_mask and class_weight is None and not avg_non_ignore
        self.loss_weight, self.loss_name = loss_weight, loss_name


@MODELS.register_module()
class MlvlSegPixelDecoder(nn.Module):
    def __init__(self, num_encoder_levels=4, in_channels=(256, 512, 1024, 2048), strides=(4, 8, 16, 32),
                 feat_channels=256, out_channels=256, num_outs=3, norm_cfg=dict(type='GN', num_groups=32),
                 act_cfg=dict(type='ReLU'), positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, normalize=True),
                 init_cfg=None):
        super().__init__()
        self.strides = list(strides)
        self.num_input_levels = len(in_channels)
        self.num_encoder_levels = num_encoder_levels
        assert self.num_input_levels == self.num_encoder_levels, \
            'the MTL configs feed every level through the shared encoder (empty FPN branch)'
        self.postional_encoding = MODELS.build(positional_encoding)
        self.level_encoding = nn.Embedding(num_encoder_levels, feat_channels)
        self.lateral_convs = nn.ModuleList()
        self.output_convs = nn.ModuleList()
        self.mask_feature = nn.Conv2d(feat_channels, out_channels, kernel_size=1, stride=1, padding=0)
        self.num_outs = num_outs
        self._pe_cache = {}

    def init_weights(self):
        nn.init.kaiming_uniform_(self.mask_feature.weight, a=1)  # caffe2_xavier_init
        nn.init.constant_(self.mask_feature.bias, 0)
        nn.init.normal_(self.level_encoding.weight, mean=0, std=1)

    def forward(self, encoder, neck_feats, backbone_feats):
        B = backbone_feats[0].shape[0]
        device = neck_feats[0].device
        inputs, shapes, refs = [], [], []
        for i in range(self.num_encoder_levels):
            level_idx = self.num_input_levels - i - 1  # low -> high resolution
            f = neck_feats[level_idx]
            h, w = f.shape[-2:]
            inputs.append(f.flatten(2).transpose(1, 2))
            shapes.append((h, w))
            refs.append(_grid_refs(h, w, self.strides[level_idx], device))
        x = torch.cat(inputs, 1)
        # level_encoding.weight[i] + positional encoding per level, concatenated (pixel_decoder.py:108-118): one launch
        # over the token-layout encoding of the level shapes (a constant)
        pkey = (tuple(shapes), str(device))
        pe_tok = self._pe_cache.get(pkey)
        if pe_tok is None:
            pe_tok = self._pe_cache[pkey] = torch.cat(
                [self.postional_encoding.unpadded(1, h, w, device).flatten(2).transpose(1, 2) for h, w in shapes], 1).contiguous()
        pos = ops.level_embed_add(None, self.level_encoding.weight, [h * w for h, w in shapes], const=pe_tok, batch=B)
        geom = LevelGeometry.get(shapes, device)
        # (a constant of the level shapes and the batch size: materialised once — every encoder layer's sampling kernel wants
        # it dense, and an expanded view would be copied by each of them)
        rkey = (tuple(shapes), B, str(device), 'ref')
        ref = self._pe_cache.get(rkey)
        if ref is None:
            ref = self._pe_cache[rkey] = torch.cat(refs, 0)[None, :, None].expand(B, -1, self.num_encoder_levels, -1).contiguous()
        # the reference passes an all-False padding mask: value.masked_fill is the identity
        memory = encoder(x, None, None, query_pos=pos, query_key_padding_mask=None, reference_points=ref,
                         **geom.kwargs())
        # one split (its backward is ONE concatenation; slicing level by level costs a full-size zero-fill + copy per level
        # and an add per level in backward)
        levels = torch.split(memory, [h * w for h, w in shapes], dim=1)
        outs = [ops.tokens_to_map(lv, hw) for lv, hw in zip(levels, shapes)]  # channels-last views
        multi_scale_features = outs[:self.num_outs]
        # 1x1 conv = MFMA GEMM on the tokens of the finest level
        h, w = shapes[-1]
        mf = ops.linear(levels[-1], self.mask_feature.weight.view(self.mask_feature.weight.shape[0], -1), self.mask_feature.bias, range_out=False)
        mask_feature = ops.tokens_to_map(mf, (h, w))
        return mask_feature, multi_scale_features


_ref_cache = {}


def _grid_refs(h, w, stride, device):
    """MlvlPointGenerator.single_level_grid_priors (offset 0.5) / (w*stride, h*stride)."""
    key = (h, w, stride, str(device))
    r = _ref_cache.get(key)
    if r is None:
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=device),
                                torch.arange(w, dtype=torch.float32, device=device), indexing='ij')
        pts = torch.stack([(xs.reshape(-1) + 0.5) * stride, (ys.reshape(-1) + 0.5) * stride], -1)
        r = pts / (torch.tensor([[w, h]], dtype=torch.float32, device=device) * stride)
        _ref_cache[key] = r
    return r


@MODELS.register_module()
class Mask2FormerHead(nn.Module):
    def __init__(self, in_channels, feat_channels, out_channels, num_classes=5, num_queries=100,
                 num_transformer_feat_level=4, scheme=1, pixel_decoder=None, enforce_decoder_input_project=False,
                 transformer_decoder=None, positional_encoding=None, ignore_index=255,
                 loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0), align_corners=False,
                 init_cfg=None):
        super().__init__()
        assert scheme == 2, 'the MTL configs use scheme=2 (queries are the output channels)'
        assert not align_corners
        self.scheme, self.num_classes, self.num_queries = scheme, num_classes, num_queries
        self.align_corners, self.ignore_index = align_corners, ignore_index
        self.num_transformer_feat_level = num_transformer_feat_level
        attn_cfgs = transformer_decoder['transformerlayers']['attn_cfgs']
        self.num_heads = (attn_cfgs[0] if isinstance(attn_cfgs, (list, tuple)) else attn_cfgs)['num_heads']
        self.num_transformer_decoder_layers = transformer_decoder['num_layers']
        pd = copy.deepcopy(dict(pixel_decoder))
        pd.update(in_channels=in_channels, feat_channels=feat_channels, out_channels=out_channels)
        self.pixel_decoder = MODELS.build(pd)
        self.transformer_decoder = MODELS.build(transformer_decoder)
        self.decoder_embed_dims = self.transformer_decoder.embed_dims
        assert self.decoder_embed_dims == feat_channels and not enforce_decoder_input_project
        self.decoder_input_projs = nn.ModuleList([nn.Identity() for _ in range(num_transformer_feat_level)])
        self.decoder_positional_encoding = MODELS.build(positional_encoding)
        self.query_embed = nn.Embedding(num_queries, feat_channels)
        self.query_feat = nn.Embedding(num_queries, feat_channels)
        self.level_embed = nn.Embedding(num_transformer_feat_level, feat_channels)
        self.mask_embed = nn.Sequential(
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, out_channels))
        self.loss_decode = MODELS.build(loss_decode)

    def init_weights(self):
        self.pixel_decoder.init_weights()
        for p in self.transformer_decoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)

    def forward_head(self, decoder_out, mask_feature, attn_mask_target_size):
        """decoder_out (B,Q,C) -> (mask_pred (B,Q,h,w), attn_mask bool (B,Q,hw_next))."""
        pn = self.transformer_decoder.post_norm
        d = ops.layer_norm(decoder_out, pn.weight, pn.bias)
        m = self.mask_embed
        e = ops.mlp(d, [(m[0].weight, m[0].bias), (m[2].weight, m[2].bias), (m[4].weight, m[4].bias)], act='relu', range_out=False)  # (read by the mask-logit products on the fp32 pipe)
        B_, C_, h_, w_ = mask_feature.shape
        # einsum('bqd,bdhw->bqhw') on the token view of the mask features (channels-last map: free) -> MFMA GEMM
        mask_pred = ops.mask_logits(e, mask_feature.permute(0, 2, 3, 1).reshape(B_, h_ * w_, C_)).view(B_, -1, h_, w_)
        attn_mask = ops.seg_attn_mask(mask_pred, attn_mask_target_size, self.num_heads)
        return mask_pred, attn_mask

    def forward(self, encoder, neck_feats, backbone_feats, img_metas, record=None):
        B = len(img_metas)
        device = neck_feats[0].device
        mask_features, memorys = self.pixel_decoder(encoder, neck_feats, backbone_feats)
        dec_in, dec_pos = [], []
        for i in range(self.num_transformer_feat_level):
            m = memorys[i]
            # (the map is a channels-last view of this level's rows of the encoder memory: read in place, batch-strided)
            dec_in.append(ops.level_embed_add(m.flatten(2).transpose(1, 2), self.level_embed.weight,
                                              [m.shape[-2] * m.shape[-1]], row0=i))
            dec_pos.append(self.decoder_positional_encoding.unpadded(B, m.shape[-2], m.shape[-1], device)
                           .flatten(2).transpose(1, 2))
        query_feat = ops.batch_param(self.query_feat.weight, B)
        query_embed = ops.batch_param(self.query_embed.weight, B)
        mask_pred, attn_mask = self.forward_head(query_feat, mask_features, memorys[0].shape[-2:])
        if record is not None:
            record['attn_masks'] = []
        nlay = self.num_transformer_decoder_layers
        # query_embed feeds both attentions of every layer: one handle each (their gradients are summed by ops.fan_out's
        # backward in 3 launches instead of 17 pairwise adds)
        n_att = len(self.transformer_decoder.layers[0].attentions)
        qe = ops.fan_out(query_embed, nlay * n_att)
        # key + key_pos of a level is the same for every layer that attends to it: formed once per level (values only; the
        # attention's backward still returns d(key)); query + query_embed leaves the previous layer's last norm
        key_sums = [torch.add(k.detach(), p) for k, p in zip(dec_in, dec_pos)] if ops.STATE.pos_sum else [None] * len(dec_in)
        q_sum = None
        for i in range(nlay):
            li = i % self.num_transformer_feat_level
            layer = self.transformer_decoder.layers[i]
            query_embed = tuple(qe[i * n_att:(i + 1) * n_att])
            if record is not None:  # in the reference's (B*heads, Q, hw) form
                record['attn_masks'].append(attn_mask.unsqueeze(1).expand(-1, self.num_heads, -1, -1).flatten(0, 1))
            nxt = qe[(i + 1) * n_att] if i + 1 < nlay else None
            query_feat = layer(query_feat, dec_in[li], dec_in[li], query_pos=query_embed, key_pos=dec_pos[li],
                               attn_masks=[attn_mask, None], query_key_padding_mask=None, key_padding_mask=None,
                               query_sum=q_sum, key_sum=key_sums[li], next_query_pos=nxt)
            if nxt is not None:
                query_feat, q_sum = query_feat
            mask_pred, attn_mask = self.forward_head(
                query_feat, mask_features, memorys[(i + 1) % self.num_transformer_feat_level].shape[-2:])
        return mask_pred  # only the last prediction is supervised (mask2former_head.py:199)

    def forward_test(self, neck_feats, backbone_feats, img_metas, shared_encoder):
        """mask2former_head.py:207-209."""
        return self(shared_encoder, neck_feats, backbone_feats, img_metas)

    def losses(self, seg_logit, seg_label):
        loss, acc = ops.upsample_ce(seg_logit, seg_label.squeeze(1), self.ignore_index)
        return {self.loss_decode.loss_name: loss * self.loss_decode.loss_weight, 'acc_seg': acc}

    def forward_train(self, neck_feats, backbone_feats, img_metas, gt_semantic_seg, shared_encoder, record=None):
        seg_logits = self.forward(shared_encoder, neck_feats, backbone_feats, img_metas, record)
        if record is not None:
            record['seg_logit'] = seg_logits
        return self.losses(seg_logits, gt_semantic_seg)

    def forward_test(self, neck_feats, backbone_feats, img_metas, shared_encoder):
        return self.forward(shared_encoder, neck_feats, backbone_feats, img_metas)

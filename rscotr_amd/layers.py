"""Neck and transformer building blocks registered under the reference's `type=` names.

Mirrors (names, parameters, argument meaning) the un-vendored layers the reference builds from
configs/multi/MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py:26-50,76-98,139-160:
mmdet `ChannelMapper`, `DetrTransformerEncoder`, `DetrTransformerDecoder`,
`SinePositionalEncoding`; mmcv `BaseTransformerLayer`, `MultiScaleDeformableAttention`,
`MultiheadAttention`, `FFN` (semantics: SURVEY.md Appendix A.2-A.6).

Layout note: the reference layers are sequence-first (L,B,C); this build keeps tokens
batch-first (B,L,C) end to end (the layout the MSDA kernel reads), so the modules take and
return batch-first tensors.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .registry import MODELS


# ------------------------------------------------------------------------------------------
@MODELS.register_module()
class ChannelMapper(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), num_outs=None, init_cfg=None):
        super().__init__()
        assert act_cfg is None and norm_cfg is not None and norm_cfg['type'] == 'GN'
        self.groups = norm_cfg['num_groups']
        self.kernel_size = kernel_size

        def conv_module(cin, k):
            m = nn.Module()
            m.conv = nn.Conv2d(cin, out_channels, k, bias=False)
            m.gn = nn.GroupNorm(self.groups, out_channels)
            return m

        self.convs = nn.ModuleList([conv_module(c, kernel_size) for c in in_channels])
        num_outs = num_outs or len(in_channels)
        self.extra_convs = None
        if num_outs > len(in_channels):
            self.extra_convs = nn.ModuleList()
            for i in range(len(in_channels), num_outs):
                self.extra_convs.append(conv_module(in_channels[-1] if i == len(in_channels) else out_channels, 3))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)

    def forward(self, inputs):
        """inputs / outputs are (B, C, H, W) maps; internally everything stays in token layout (the 1x1
        convolutions are MFMA GEMMs on the token matrix, the 3x3/s2 one a gather + GEMM, GroupNorm on
        tokens) and the outputs are channels-last views of the token tensors."""
        assert len(inputs) == len(self.convs) and self.kernel_size == 1
        toks = [ops.map_to_tokens(x) for x in inputs]
        outs = []
        for (t, hw), m in zip(toks, self.convs):
            y = ops.linear(t, m.conv.weight.view(m.conv.weight.shape[0], -1), None, range_out=False)  # (read by the GroupNorm)
            outs.append((ops.group_norm_tokens(y, self.groups, m.gn.weight, m.gn.bias), hw))
        if self.extra_convs:
            for i, m in enumerate(self.extra_convs):
                src, hw = toks[-1] if i == 0 else outs[-1]
                y, hw2 = ops.conv3x3s2_tokens(src, hw, m.conv.weight)
                outs.append((ops.group_norm_tokens(y, self.groups, m.gn.weight, m.gn.bias), hw2))
        return tuple(ops.tokens_to_map(t, hw) for t, hw in outs)


# ------------------------------------------------------------------------------------------
@MODELS.register_module()
class SinePositionalEncoding(nn.Module):
    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6,
                 offset=0., init_cfg=None):
        super().__init__()
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset
        self._cache = {}

    def forward(self, mask):
        """mask (B,H,W) bool, True = padded -> (B, 2*num_feats, H, W)."""
        mask = mask.to(torch.int)
        not_mask = 1 - mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = (y_embed + self.offset) / (y_embed[:, -1:, :] + self.eps) * self.scale
            x_embed = (x_embed + self.offset) / (x_embed[:, :, -1:] + self.eps) * self.scale
        dim_t = torch.arange(self.num_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        B, H, W = mask.size()
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)

    def unpadded(self, B, H, W, device):
        """Encoding of an all-valid (B,H,W) mask, cached per shape (it is a constant)."""
        key = (H, W, str(device))
        pe = self._cache.get(key)
        if pe is None:
            pe = self.forward(torch.zeros((1, H, W), dtype=torch.bool, device=device))
            self._cache[key] = pe
        return pe.expand(B, -1, -1, -1)


# ------------------------------------------------------------------------------------------
@MODELS.register_module()
class FFN(nn.Module):
    """mmcv FFN: x + Linear(ReLU(Linear(x))), names layers.0.0 / layers.1."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True),
                 ffn_drop=0., dropout_layer=None, add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        assert num_fcs == 2 and ffn_drop == 0. and act_cfg['type'] == 'ReLU'
        self.add_identity = add_identity
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True), nn.Identity()),
            nn.Linear(feedforward_channels, embed_dims), nn.Identity())

    def forward(self, x, identity=None):
        idt = None if not self.add_identity else (x if identity is None else identity)
        return ops.mlp(x, [(self.layers[0][0].weight, self.layers[0][0].bias),
                           (self.layers[1].weight, self.layers[1].bias)], act='relu', identity=idt)


@MODELS.register_module()
class MultiScaleDeformableAttention(nn.Module):
    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1,
                 batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__()
        assert dropout == 0.0, 'the MTL configs set dropout=0.0'
        assert embed_dims % num_heads == 0
        self.embed_dims, self.num_heads, self.num_levels, self.num_points = embed_dims, num_heads, num_levels, num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2).repeat(
            1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, offset_norm=None, query_sum=None,
                **kwargs):
        """Batch-first: query (B,Nq,C), value (B,Nk,C); reference_points (B,Nq,L,2|4);
        spatial_shapes (L,2) int64 device tensor; offset_norm (L,2) float (W_l,H_l); query_sum = query + query_pos where
        the producer of `query` already formed it (ops.layer_norm_sum)."""
        if identity is None:
            identity = query
        # projections, softmax / location arithmetic, sampling kernel, output projection + identity: one autograd node
        # (ops._MSDAAttn); `value is None` / `value is query` = self-attention over the token map
        return ops.msda_attention(
            query, query_pos, value, identity, key_padding_mask, reference_points, spatial_shapes, level_start_index,
            offset_norm, self.num_heads, self.num_levels, self.num_points,
            self.sampling_offsets.weight, self.sampling_offsets.bias, self.attention_weights.weight,
            self.attention_weights.bias, self.value_proj.weight, self.value_proj.bias, self.output_proj.weight,
            self.output_proj.bias, q_sum=query_sum if query_pos is not None else None)


@MODELS.register_module()
class MultiheadAttention(nn.Module):
    """mmcv MultiheadAttention wrapper around torch.nn.MultiheadAttention;
    the packed in_proj/out_proj parameters live under `.attn.` as in the reference state dict."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., dropout_layer=None, init_cfg=None,
                 batch_first=False, dropout=None, **kwargs):
        super().__init__()
        assert attn_drop == 0. and proj_drop == 0. and not dropout
        self.embed_dims, self.num_heads = embed_dims, num_heads
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, 0.0)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, query_sum=None, key_sum=None, **kwargs):
        if key_padding_mask is not None:
            raise NotImplementedError('key_padding_mask is never set for dense attention on this path')
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        a = self.attn
        mode = kwargs.get('attn_mask_mode')
        # the positional adds, the projections, the attention core, the output projection and the identity are one autograd
        # node (ops._MHA): what meets at `query` in backward is merged in GEMM epilogues
        return ops.mha(query, key, value, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                       self.num_heads, attn_mask, identity=identity, mask_mode=mode, q_pos=query_pos, k_pos=key_pos,
                       q_sum=query_sum if query_pos is not None else None,
                       k_sum=key_sum if (key_pos is not None and key is not query) else None)


@MODELS.register_module()
class BaseTransformerLayer(nn.Module):
    def __init__(self, attn_cfgs=None, ffn_cfgs=None, operation_order=None, norm_cfg=dict(type='LN'),
                 init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        assert set(operation_order) <= {'self_attn', 'norm', 'ffn', 'cross_attn'}
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [dict(attn_cfgs) for _ in range(num_attn)]
        assert len(attn_cfgs) == num_attn
        self.operation_order = tuple(operation_order)
        self.pre_norm = operation_order[0] == 'norm'
        assert not self.pre_norm, 'the MTL configs are post-norm'
        self.attentions = nn.ModuleList([MODELS.build(c) for c in attn_cfgs])
        self.embed_dims = self.attentions[0].embed_dims
        ffn_cfgs = dict(ffn_cfgs or {})
        ffn_cfgs.setdefault('type', 'FFN')
        ffn_cfgs.setdefault('embed_dims', self.embed_dims)
        self.ffns = nn.ModuleList([MODELS.build(ffn_cfgs) for _ in range(operation_order.count('ffn'))])
        self.norms = nn.ModuleList([nn.LayerNorm(self.embed_dims) for _ in range(operation_order.count('norm'))])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, query_sum=None, key_sum=None, next_query_pos=None,
                **kwargs):
        """query_sum: query + (the first attention's query_pos) where the caller already has it; key_sum: key + key_pos
        likewise (cross-attention); next_query_pos: the positional embedding the NEXT consumer of this layer's output adds
        to it — the layer's last `norm` then emits that sum from its own launch and the result is (output, output +
        next_query_pos).  Sums are values only (no gradient of their own): every `norm` that is followed by an attention
        emits `norm(x) + query_pos` the same way instead of the attention wrapper launching an element-wise add."""
        norm_i = attn_i = ffn_i = 0
        sums = ops.STATE.pos_sum
        q_sum = query_sum if sums else None
        out_sum = None
        order = self.operation_order
        if attn_masks is None:
            attn_masks = [None] * len(self.attentions)
        elif torch.is_tensor(attn_masks):
            attn_masks = [attn_masks for _ in self.attentions]
        qp_all = query_pos

        def pos_of(i):
            # a tuple / list of query_pos: one handle per attention of the layer (ops.fan_out: the gradients of all
            # consumers of a shared positional embedding are then summed by one launch)
            return qp_all[i] if isinstance(qp_all, (tuple, list)) else qp_all

        for j, op in enumerate(order):
            if op in ('self_attn', 'cross_attn'):
                query_pos = pos_of(attn_i)
            if op == 'self_attn':
                query = self.attentions[attn_i](query, query, query, None, query_pos=query_pos, key_pos=query_pos,
                                                attn_mask=attn_masks[attn_i],
                                                key_padding_mask=query_key_padding_mask, query_sum=q_sum, **kwargs)
                attn_i += 1
                q_sum = None
            elif op == 'norm':
                n = self.norms[norm_i]
                last = j == len(order) - 1
                nxt = order[j + 1] if not last else None
                add = None
                if sums:
                    if nxt in ('self_attn', 'cross_attn'):
                        add = pos_of(attn_i)
                    elif last:
                        add = next_query_pos
                if torch.is_tensor(add) and (add.shape == query.shape or add.shape[1:] == query.shape[1:]):
                    query, s_ = ops.layer_norm_sum(query, n.weight, n.bias, add)
                    if last:
                        out_sum = s_
                    else:
                        q_sum = s_
                else:
                    query = ops.layer_norm(query, n.weight, n.bias)
                norm_i += 1
            elif op == 'cross_attn':
                query = self.attentions[attn_i](query, key, value, None, query_pos=query_pos, key_pos=key_pos,
                                                attn_mask=attn_masks[attn_i], key_padding_mask=key_padding_mask,
                                                query_sum=q_sum, key_sum=key_sum if sums else None, **kwargs)
                attn_i += 1
                q_sum = None
            else:
                query = self.ffns[ffn_i](query)
                ffn_i += 1
                q_sum = None
        if next_query_pos is not None:
            return query, out_sum
        return query


class TransformerLayerSequence(nn.Module):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__()
        if isinstance(transformerlayers, dict):
            transformerlayers = [dict(transformerlayers) for _ in range(num_layers)]
        self.num_layers = num_layers
        self.layers = nn.ModuleList([MODELS.build(c) for c in transformerlayers])
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm

    def forward(self, query, key=None, value=None, **kwargs):
        # the positional embedding is shared by all layers: one handle per layer, so that its gradients meet in one launch
        qp = kwargs.get('query_pos')
        qps = ops.fan_out(qp, len(self.layers)) if torch.is_tensor(qp) else None
        q_sum = None
        for i, layer in enumerate(self.layers):
            if qps is not None:
                kwargs['query_pos'] = qps[i]
            if qps is not None and i + 1 < len(self.layers):
                # the next layer's first attention adds the same embedding to this layer's output: that sum leaves this
                # layer's last norm
                query, q_sum = layer(query, key, value, query_sum=q_sum, next_query_pos=qps[i + 1], **kwargs)
            else:
                query = layer(query, key, value, query_sum=q_sum, **kwargs)
        return query


@MODELS.register_module()
class DetrTransformerEncoder(TransformerLayerSequence):
    """Post-norm mode has no final norm (mmdet DetrTransformerEncoder)."""

    def __init__(self, *args, post_norm_cfg=dict(type='LN'), **kwargs):
        super().__init__(*args, **kwargs)
        self.post_norm = None


@MODELS.register_module()
class DetrTransformerDecoder(TransformerLayerSequence):
    def __init__(self, *args, post_norm_cfg=dict(type='LN'), return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.post_norm = nn.LayerNorm(self.embed_dims) if post_norm_cfg is not None else None


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class LevelGeometry:
    """Per-call constants of one multi-level token layout, built once on the host from python
    ints (no device sync): spatial_shapes / level_start_index device tensors for the MSDA
    kernel and the (W_l,H_l) offset normaliser."""
    _cache = {}

    def __init__(self, shapes, device):
        self.shapes = [tuple(int(v) for v in s) for s in shapes]
        self.spatial_shapes = torch.tensor(self.shapes, dtype=torch.long, device=device)
        ops.msda_register_shapes(self.spatial_shapes, self.shapes)
        starts, s = [], 0
        for h, w in self.shapes:
            starts.append(s)
            s += h * w
        self.starts = starts
        self.num_tokens = s
        self.level_start_index = torch.tensor(starts, dtype=torch.long, device=device)
        self.offset_norm = torch.tensor([(w, h) for h, w in self.shapes], dtype=torch.float32, device=device)

    @classmethod
    def get(cls, shapes, device):
        key = (tuple(tuple(int(v) for v in s) for s in shapes), str(device))
        g = cls._cache.get(key)
        if g is None:
            g = cls._cache[key] = cls(shapes, device)
        return g

    def kwargs(self):
        return dict(spatial_shapes=self.spatial_shapes, level_start_index=self.level_start_index,
                    offset_norm=self.offset_norm)

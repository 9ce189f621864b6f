"""Swin Transformer backbone (`type='SwinTransformer'`).

Host-side mirror of mmdet 2.25.1 `SwinTransformer` as configured at
configs/multi/MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py:9-25 and called from
models/multi/multitask_learner.py:83.  Parameter names follow SURVEY.md Appendix A.8 so
reference checkpoints map one-to-one; the compute goes through rscotr_amd.ops.
"""
import torch
import torch.nn as nn

from . import ops
from .registry import MODELS


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def rel_pos_index(ws):
    c = torch.arange(ws)
    yy, xx = torch.meshgrid(c, c, indexing='ij')
    y, x = yy.reshape(-1), xx.reshape(-1)
    return (y[:, None] - y[None, :] + ws - 1) * (2 * ws - 1) + (x[:, None] - x[None, :] + ws - 1)


class WindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size, qkv_bias=True):
        super().__init__()
        self.embed_dims, self.num_heads, self.window_size = embed_dims, num_heads, window_size
        ws = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        self.register_buffer('relative_position_index', rel_pos_index(ws))
        self.qkv = nn.Linear(embed_dims, embed_dims * 3, bias=qkv_bias)
        self.proj = nn.Linear(embed_dims, embed_dims)
        trunc_normal_(self.relative_position_bias_table, std=0.02)


class ShiftWindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size, shift_size, qkv_bias, drop_path):
        super().__init__()
        self.window_size, self.shift_size, self.drop_path = window_size, shift_size, drop_path
        self.w_msa = WindowMSA(embed_dims, num_heads, window_size, qkv_bias)

    def forward(self, x, hw, identity=None, out_scale=None):
        w = self.w_msa
        return ops.swin_window_attention(
            x, hw, w.qkv.weight, w.qkv.bias, w.relative_position_bias_table,
            w.relative_position_index, w.proj.weight, w.proj.bias, w.num_heads,
            self.window_size, self.shift_size, identity=identity, out_scale=out_scale)


class SwinFFN(nn.Module):
    """mmcv FFN(num_fcs=2, GELU): layers = Sequential(Sequential(Linear, GELU, Dropout), Linear, Dropout)."""

    def __init__(self, embed_dims, hidden):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, hidden), nn.GELU(), nn.Identity()),
            nn.Linear(hidden, embed_dims), nn.Identity())

    def forward(self, x, identity=None, out_scale=None):
        return ops.mlp(x, [(self.layers[0][0].weight, self.layers[0][0].bias),
                           (self.layers[1].weight, self.layers[1].bias)], act='gelu', identity=identity,
                       out_scale=out_scale)


class SwinBlock(nn.Module):
    def __init__(self, embed_dims, num_heads, hidden, window_size, shift, qkv_bias, drop_path):
        super().__init__()
        self.drop_path = drop_path
        self.norm1 = nn.LayerNorm(embed_dims)
        self.attn = ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0,
                                   qkv_bias, drop_path)
        self.norm2 = nn.LayerNorm(embed_dims)
        self.ffn = SwinFFN(embed_dims, hidden)

    def forward(self, x, hw, scale_attn=None, scale_ffn=None):
        """x + DropPath(attn(LN1 x)); then x + DropPath(ffn(LN2 x)).  scale_* (B,) = keep / keep_prob of mmcv's
        drop_path (None: no stochastic depth): residual add and scaling ride the proj / fc2 GEMM epilogues."""
        # layer_norm_fork hands x back as the residual: its gradient is added inside the LayerNorm backward kernel
        # (lazy: the norm's output has one reader, the qkv Linear, which takes the norm as its prologue where that launch exists)
        y, x = ops.layer_norm_fork(x, self.norm1.weight, self.norm1.bias, lazy=True)
        x = self.attn(y, hw, identity=x, out_scale=scale_attn)
        # (lazy: the norm's launch is left to the MLP call, its only reader, which folds it into the fused kernel where that exists)
        y, x = ops.layer_norm_fork(x, self.norm2.weight, self.norm2.bias, lazy=True)
        return self.ffn(y, identity=x, out_scale=scale_ffn)


class PatchMerging(nn.Module):
    """mmcv PatchMerging: nn.Unfold(2, stride 2) (channel-major order c*4+kh*2+kw) -> LN(4C) ->
    Linear(4C, 2C, bias=False)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.norm = nn.LayerNorm(4 * in_channels)
        self.reduction = nn.Linear(4 * in_channels, out_channels, bias=False)

    def forward(self, x, hw):
        if ops.STATE.merge_norm and x.shape[-1] <= 512:  # unfold done by the norm's own loads / stores
            y, hw2 = ops.patch_merge_norm(x, hw, self.norm.weight, self.norm.bias)
        else:
            y, hw2 = ops.patch_merge_gather(x, hw)
            y = ops.layer_norm(y, self.norm.weight, self.norm.bias)
        return ops.linear(y, self.reduction.weight, None, range_out=False), hw2  # (read by the next stage's norm)


class SwinBlockSequence(nn.Module):
    def __init__(self, embed_dims, num_heads, hidden, depth, window_size, qkv_bias, drop_paths, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinBlock(embed_dims, num_heads, hidden, window_size, i % 2 == 1, qkv_bias, drop_paths[i])
            for i in range(depth)])
        self.downsample = downsample


class PatchEmbed(nn.Module):
    def __init__(self, in_channels, embed_dims, patch_size):
        super().__init__()
        self.patch_size = patch_size
        self.projection = nn.Conv2d(in_channels, embed_dims, patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dims)

    def forward(self, img):
        x, hw = ops.patch_embed(img, self.projection.weight, self.projection.bias, self.patch_size)
        return ops.layer_norm(x, self.norm.weight, self.norm.bias), hw


@MODELS.register_module()
class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, in_channels=3, embed_dims=96, patch_size=4, window_size=7,
                 mlp_ratio=4, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), strides=(4, 2, 2, 2),
                 out_indices=(0, 1, 2, 3), qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.1, use_abs_pos_embed=False, act_cfg=None,
                 norm_cfg=None, with_cp=False, pretrained=None, convert_weights=False,
                 frozen_stages=-1, init_cfg=None):
        super().__init__()
        assert not use_abs_pos_embed and patch_norm and qk_scale is None
        assert drop_rate == 0. and attn_drop_rate == 0.
        self.out_indices = tuple(out_indices)
        self.depths = tuple(depths)
        self.convert_weights = convert_weights
        self.init_cfg = init_cfg
        self.patch_embed = PatchEmbed(in_channels, embed_dims, patch_size)
        total = sum(depths)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, total)]
        self.drop_path_rates = dpr
        self.stages = nn.ModuleList()
        c = embed_dims
        self.num_features = []
        for i, d in enumerate(depths):
            down = PatchMerging(c, 2 * c) if i < len(depths) - 1 else None
            self.stages.append(SwinBlockSequence(c, num_heads[i], int(mlp_ratio * c), d, window_size, qkv_bias,
                                                 dpr[sum(depths[:i]):sum(depths[:i + 1])], down))
            self.num_features.append(c)
            if down is not None:
                c = 2 * c
        for i in self.out_indices:
            self.add_module(f'norm{i}', nn.LayerNorm(self.num_features[i]))

    def init_weights(self):
        """mmdet SwinTransformer.init_weights without a checkpoint: trunc_normal(0.02) Linear
        weights, zero bias, LayerNorm (1, 0)."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def num_drop_draws(self):
        return 2 * sum(self.depths)

    def forward(self, img, drop_keep=None):
        """img (B,3,H,W) -> tuple of (B,C_i,H_i,W_i). drop_keep: (2*blocks, B) float 0/1 keep
        flags (train mode DropPath draws) or None for no stochastic depth."""
        x, hw = self.patch_embed(img)
        outs = []
        blk = 0
        scales = None
        if drop_keep is not None:  # mmcv drop_path: y / keep_prob * floor(keep_prob + U), all blocks at once
            kp = getattr(self, '_keep_prob_t', None)
            if kp is None or kp.device != drop_keep.device:
                kp = self._keep_prob_t = (1.0 - torch.tensor(self.drop_path_rates, device=drop_keep.device)
                                          .repeat_interleave(2))[:, None]
            scales = drop_keep / kp
        for i, stage in enumerate(self.stages):
            for b in stage.blocks:
                sa = None if (scales is None or b.drop_path == 0.0) else scales[2 * blk]
                sf = None if (scales is None or b.drop_path == 0.0) else scales[2 * blk + 1]
                x = b(x, hw, sa, sf)
                blk += 1
            if i in self.out_indices:
                n = getattr(self, f'norm{i}')
                if stage.downsample is not None:
                    # the stage output feeds its out-norm AND the next stage: forked at the norm, whose backward kernel adds
                    # the downsample branch's gradient on its way out (no element-wise add by autograd)
                    o, x = ops.layer_norm_fork(x, n.weight, n.bias)
                else:
                    o = ops.layer_norm(x, n.weight, n.bias)
                outs.append(ops.tokens_to_map(o, hw))
            if stage.downsample is not None:
                x, hw = stage.downsample(x, hw)
        return tuple(outs)

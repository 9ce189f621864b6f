"""Checkpoint interop.

`swin_converter`: official Swin-Transformer checkpoint keys / PatchMerging orderings -> the mmdet SwinTransformer layout
this repo's backbone uses (what `convert_weights=True` does in the reference, cfg ...potsdam.py:24; mmdet
`models/utils/ckpt_convert.py::swin_converter`, un-vendored).  The official PatchMerging concatenates the four 2x2
neighbours as [x(0,0), x(1,0), x(0,1), x(1,1)] (neighbour-major, channel-minor); mmdet's uses nn.Unfold, whose feature
index is c*4 + kh*2 + kw (channel-major) — so the `reduction` weight columns and the `norm` parameters of every
downsample layer are permuted.  `load_pretrained_backbone` applies it and loads non-strictly;
`MTL.load_task_pretrain` (rscotr_amd/mtl.py) handles the task checkpoints (`multitask_learner.py:308-353`).

`save_checkpoint` / `load_checkpoint` / `resume` write and read the mmcv 1.x checkpoint layout the reference's runner
produces (`dict(meta=..., state_dict=..., optimizer=...)`, mmcv `CheckpointHook`, cfg ...potsdam.py:218;
`mtl/apis/train.py:109-118` resume): parameter names are the reference's, the optimizer entry is
torch.optim.AdamW's state layout, so a reference `.pth` loads here and a checkpoint written here loads there.
"""
from collections import OrderedDict

import torch


def _correct_unfold_reduction_order(x):
    out_channel, in_channel = x.shape
    x = x.reshape(out_channel, 4, in_channel // 4)
    return x[:, [0, 2, 1, 3], :].transpose(1, 2).reshape(out_channel, in_channel)


def _correct_unfold_norm_order(x):
    in_channel = x.shape[0]
    x = x.reshape(4, in_channel // 4)
    return x[[0, 2, 1, 3], :].transpose(0, 1).reshape(in_channel)


def swin_converter(ckpt, prefix='backbone.'):
    new_ckpt = OrderedDict()
    for k, v in ckpt.items():
        if k.startswith('head'):
            continue
        new_v = v
        if k.startswith('layers'):
            if 'attn.' in k:
                new_k = k.replace('attn.', 'attn.w_msa.')
            elif 'mlp.' in k:
                if 'mlp.fc1.' in k:
                    new_k = k.replace('mlp.fc1.', 'ffn.layers.0.0.')
                elif 'mlp.fc2.' in k:
                    new_k = k.replace('mlp.fc2.', 'ffn.layers.1.')
                else:
                    new_k = k.replace('mlp.', 'ffn.')
            elif 'downsample' in k:
                new_k = k
                if 'reduction.' in k:
                    new_v = _correct_unfold_reduction_order(v)
                elif 'norm.' in k:
                    new_v = _correct_unfold_norm_order(v)
            else:
                new_k = k
            new_k = new_k.replace('layers', 'stages', 1)
        elif k.startswith('patch_embed'):
            new_k = k.replace('proj', 'projection') if 'proj' in k else k
        else:
            new_k = k
        new_ckpt[prefix + new_k] = new_v
    return new_ckpt


def load_pretrained_backbone(model, path, convert_weights=True, map_location='cpu'):
    """Load an (official or mmdet-layout) Swin checkpoint into `model.backbone`; returns torch's load report."""
    sd = torch.load(path, map_location=map_location, weights_only=True)
    for key in ('state_dict', 'model'):
        if isinstance(sd, dict) and key in sd:
            sd = sd[key]
    sd = swin_converter(sd) if convert_weights else OrderedDict(sd)
    skip = [k for k in sd if k.endswith(('relative_position_index', 'attn_mask'))]  # buffers recomputed here
    for k in skip:
        del sd[k]
    from . import ops
    ops.WPLANES.bump()  # (also when `model` is a bare backbone outside an MTL: planes and parameter range words are stale)
    return model.load_state_dict(sd, strict=False)


def _strip_module_prefix(sd):
    """mmcv load_state_dict: checkpoints saved from a (Distributed)DataParallel wrapper carry a `module.` prefix."""
    if sd and all(k.startswith('module.') for k in sd):
        return OrderedDict((k[7:], v) for k, v in sd.items())
    return sd


def save_checkpoint(path, model, optimizer=None, meta=None):
    """mmcv.runner.save_checkpoint layout: weights on the CPU, `meta` (iter, CLASSES, ...) alongside."""
    ckpt = dict(meta=dict(meta or {}),
                state_dict=OrderedDict((k, v.detach().cpu().clone()) for k, v in model.state_dict().items()))
    if getattr(model, 'CLASSES', None) is not None:
        ckpt['meta'].setdefault('CLASSES', model.CLASSES)
    if optimizer is not None:
        osd = optimizer.state_dict()
        for st in osd['state'].values():
            for k, v in st.items():
                if torch.is_tensor(v):
                    st[k] = v.detach().cpu()
        ckpt['optimizer'] = osd
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(model, path, strict=False, map_location='cpu', weights_only=True):
    """mmcv.runner.load_checkpoint: accepts a bare state dict or dict(state_dict=...), strips a `module.` prefix,
    copies INTO the existing parameter storage (the parameters are views of the optimizer's flat arena).  Returns
    (checkpoint, torch's missing / unexpected key report).  `weights_only=True` (tensors and plain containers only) is the
    default: pass False only for a trusted file whose `meta` holds arbitrary pickled objects."""
    ckpt = torch.load(path, map_location=map_location, weights_only=weights_only) if isinstance(path, str) else path
    sd = ckpt['state_dict'] if isinstance(ckpt, dict) and 'state_dict' in ckpt else ckpt
    report = model.load_state_dict(_strip_module_prefix(sd), strict=strict)
    from . import ops
    ops.WPLANES.bump()  # parameters changed in place: pre-split planes are stale
    if isinstance(ckpt, dict) and 'CLASSES' in ckpt.get('meta', {}):
        model.CLASSES = ckpt['meta']['CLASSES']
    return ckpt, report


def resume(runner, path, map_location='cpu'):
    """IterBasedRunner.resume: weights, optimizer moments / step counts and the iteration counter."""
    ckpt, report = load_checkpoint(runner.model, path, strict=True, map_location=map_location)
    if 'optimizer' in ckpt:
        runner.optimizer.load_state_dict(ckpt['optimizer'])
    runner.iter = int(ckpt.get('meta', {}).get('iter', 0))
    return ckpt

"""`type='MTL'` — the multi-task learner (models/multi/multitask_learner.py:35-353).

Same constructor arguments, `forward` / `train_step` / `val_step` contract, loss-key naming and
task weighting as the reference.  Deliberate, result-preserving differences:
  * `_parse_losses` packs every log scalar into ONE vector: one all-reduce (distributed) and one
    device->host copy per step instead of K+1 all-reduces and K `.item()` syncs
    (multitask_learner.py:289-304);
  * the cls step skips the neck when the head is SlvlClsHead (its output is discarded by that head,
    multitask_learner.py:122 / SURVEY.md A.7(5)) — gradients are identical because the neck gets
    none on cls steps in the reference either;
  * stochastic draws (DropPath, Mixup/CutMix, CDN noise) can be injected through `rnd=` so the
    oracle sees the same randomness.
"""
from collections import OrderedDict
from collections.abc import Mapping

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .cls_head import Augments
from .layers import MultiScaleDeformableAttention
from .registry import MODELS, build_backbone, build_head, build_neck, build_transformer_layer_sequence

supported_tasks = ('cls', 'det', 'seg')


def bbox2result(bboxes, labels, num_classes):
    """mmdet.core.bbox2result: (n,5) boxes + (n,) labels -> list of per-class (k,5) float32 arrays."""
    import numpy as np
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    bboxes, labels = bboxes.detach().cpu().numpy(), labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


class LazyLogVars(Mapping):
    """`log_vars` of a step: an ordered name -> python float mapping whose values live in ONE packed
    device vector until somebody reads them.  The reference calls `.item()` on every scalar right after
    the forward pass (multitask_learner.py:299-304), which stalls the host in the middle of the step; the
    values are only consumed by the logger every `interval` iterations, so the copy is deferred to the
    first access (one device->host copy for all scalars)."""

    def __init__(self, names, packed):
        self._all = list(names)                      # may repeat a key (cls: 'loss' twice)
        self._names = list(dict.fromkeys(self._all))  # dict semantics: first position, last value
        self._packed, self._vals = packed, None

    def _get(self):
        if self._vals is None:
            self._vals = OrderedDict(zip(self._all, self._packed.tolist()))
            self._packed = None
        return self._vals

    def prefixed(self, prefix):
        assert self._vals is None
        return LazyLogVars([f'{prefix}.{n}' for n in self._all], self._packed)

    def all_reduced(self):
        """Rank-averaged values (multitask_learner.py:299-304): ONE all-reduce of the packed vector, still no host sync."""
        assert self._vals is None
        from .dist import mean_over_ranks
        packed = mean_over_ranks(self._packed.float().clone().contiguous())
        return LazyLogVars(self._all, packed)

    def scaled(self, weight):
        assert self._vals is None
        return LazyLogVars(self._all, self._packed * weight)

    def __getitem__(self, k):
        return self._get()[k]

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)


def add_prefix(inputs, prefix):
    if isinstance(inputs, LazyLogVars):
        return inputs.prefixed(prefix)
    return OrderedDict((f'{prefix}.{k}', v) for k, v in inputs.items())


def _params_rewritten(*_args, **_kwargs):
    ops.WPLANES.bump()


@MODELS.register_module()
class MTL(nn.Module):
    PALETTE = None

    def __init__(self, backbone, neck, shared_encoder, cls_head=None, bbox_head=None, seg_head=None,
                 task_weight=None, train_cfg=None, test_cfg=None, init_cfg=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck)
        self.shared_encoder = build_transformer_layer_sequence(shared_encoder)
        self.task_weight = dict(cls=1, det=1, seg=1)
        if task_weight is not None:
            assert isinstance(task_weight, dict)
            self.task_weight.update(task_weight)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.cls_augments = None
        cls_augments_cfg = train_cfg['cls'].get('augments', None)
        if cls_augments_cfg is not None:
            self.cls_augments = Augments(cls_augments_cfg)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg['det'])
        bbox_head.update(test_cfg=test_cfg['det'])
        self.task_pretrain = self.train_cfg.get('task_pretrain', None)
        self.cls_head = build_head(cls_head, 'mmcls')
        self.bbox_head = build_head(bbox_head, 'mmdet')
        self.seg_head = build_head(seg_head, 'mmseg')
        self.CLASSES = None
        # nn.Module.load_state_dict recurses through _load_from_state_dict and never calls a child's load_state_dict(): a load
        # into a submodule (the backbone's pre-training) or through a wrapper would leave the pre-split weight planes and the
        # parameters' value-range words stale (a stale-small word overflows the fp16 planes: ADVICE r5) — every module says so itself
        for m in self.modules():
            m._register_load_state_dict_pre_hook(_params_rewritten)

    # -------------------------------------------------------------------------------------
    def init_weights(self):
        for m in (self.backbone, self.neck, self.cls_head, self.bbox_head, self.seg_head):
            if hasattr(m, 'init_weights'):
                m.init_weights()
        # the encoder keeps torch's default Linear init (no init_cfg in the reference); MSDA re-init:
        for layer in self.shared_encoder.layers:
            for attn in layer.attentions:
                if isinstance(attn, MultiScaleDeformableAttention):
                    attn.init_weights()
        ops.WPLANES.bump()  # (in-place re-initialisation: planes and range words of the parameters are stale)

    def extract_feat(self, img, drop_keep=None, with_neck=True):
        backbone_feature = self.backbone(img, drop_keep)
        neck_feature = self.neck(backbone_feature[-3:]) if with_neck else None
        return neck_feature, backbone_feature

    def _drop_keep(self, B, device, rnd):
        if rnd is not None and 'drop_keep' in rnd:
            dk = rnd['drop_keep']
            return None if dk is None else dk.to(device)
        if not self.training:
            return None
        if max(self.backbone.drop_path_rates) == 0.0:
            return None
        keep = getattr(self, '_keep_prob', None)
        if keep is None or keep.device != device:  # built once (a host->device copy cannot be captured)
            rates = torch.tensor(self.backbone.drop_path_rates, device=device).repeat_interleave(2)
            keep = self._keep_prob = (1 - rates)[:, None]
        return torch.floor(keep + torch.rand(keep.shape[0], B, device=device))

    # -------------------------------------------------------------------------------------
    def forward_train(self, task, *args, **kwargs):
        assert task in supported_tasks
        return getattr(self, f'forward_train_{task}')(*args, **kwargs)

    def forward_train_cls(self, img, gt_label, img_metas=None, rnd=None, record=None, **kwargs):
        if self.cls_augments is None:
            raise AttributeError("'MTL' object has no attribute 'cls_augments'")  # reference quirk A.7(15)
        if rnd is not None and 'cls_aug_static' in rnd:  # shape-static form (graph capture)
            img, gt_label = self.cls_augments.apply_static(img, gt_label, rnd['cls_aug_static'])
        else:
            img, gt_label = self.cls_augments(img, gt_label, None if rnd is None else rnd.get('cls_aug'))
        neck_feature, backbone_feature = self.extract_feat(img, self._drop_keep(img.shape[0], img.device, rnd),
                                                           with_neck=getattr(self.cls_head, 'needs_neck', True))
        if record is not None:
            record['backbone_feats'] = backbone_feature
            if neck_feature is not None:
                record['neck_feats'] = neck_feature
        losses = dict()
        losses.update(self.cls_head.forward_train(neck_feature, backbone_feature, gt_label, self.shared_encoder))
        return losses

    def forward_train_det(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, rnd=None, record=None,
                          static=None, gt_bboxes_host=None, gt_labels_host=None, det_norms_r_host=None):
        batch_input_shape = tuple(img[0].size()[-2:])
        for img_meta in img_metas:
            img_meta['batch_input_shape'] = batch_input_shape
        x, bf = self.extract_feat(img, self._drop_keep(img.shape[0], img.device, rnd))
        if record is not None:
            record['backbone_feats'], record['neck_feats'] = bf, x
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore, self.shared_encoder,
                                            rnd=None if rnd is None else rnd.get('cdn'), record=record, static=static,
                                            gt_host=None if gt_bboxes_host is None or gt_labels_host is None
                                            else (gt_bboxes_host, gt_labels_host), norms_r_host=det_norms_r_host)

    def forward_train_seg(self, img, img_metas, gt_semantic_seg, rnd=None, record=None):
        neck_feature, backbone_feature = self.extract_feat(img, self._drop_keep(img.shape[0], img.device, rnd))
        if record is not None:
            record['backbone_feats'], record['neck_feats'] = backbone_feature, neck_feature
        loss_decode = self.seg_head.forward_train(neck_feature, backbone_feature, img_metas, gt_semantic_seg,
                                                  self.shared_encoder, record=record)
        losses = dict()
        losses.update(add_prefix(loss_decode, 'seg'))
        return losses

    # ---- inference (multitask_learner.py:91-227): same kernels, no losses ---------------------------
    def forward_test(self, task, img, img_metas, *args, **kwargs):
        if isinstance(task, list):
            task = list(set(task))
            if len(task) != 1:
                raise NotImplementedError('The current implementation only support same task in a batch')
            task = task[0]
        if isinstance(img, list):
            if len(img) != 1:
                raise NotImplementedError('The current implementation does not support TTA ')
            img = img[0]
        if isinstance(img_metas[0], list):
            img_metas = img_metas[0]
        return self.simple_test(task, img, img_metas, *args, **kwargs)

    def simple_test(self, task, *args, **kwargs):
        assert task in supported_tasks
        return getattr(self, f'simple_test_{task}')(*args, **kwargs)

    def simple_test_cls(self, img, img_metas=None, **kwargs):
        neck_feature, backbone_feature = self.extract_feat(img, with_neck=getattr(self.cls_head, 'needs_neck', True))
        return self.cls_head.simple_test(neck_feature, backbone_feature, shared_encoder=self.shared_encoder, **kwargs)

    def simple_test_det(self, img, img_metas, rescale=False):
        for m in img_metas:
            m['batch_input_shape'] = tuple(img.size()[-2:])
        feat = self.extract_feat(img)[0]
        results_list = self.bbox_head.simple_test(feat, img_metas, rescale=rescale, shared_encoder=self.shared_encoder)
        return [bbox2result(b, l, self.bbox_head.num_classes) for b, l in results_list]

    def whole_inference_seg(self, img, img_meta, rescale):
        neck_feature, backbone_feature = self.extract_feat(img)
        seg_logit = self.seg_head.forward_test(neck_feature, backbone_feature, img_meta, self.shared_encoder)
        seg_logit = torch.nn.functional.interpolate(seg_logit, size=img.shape[2:], mode='bilinear',
                                                    align_corners=self.seg_head.align_corners)
        if rescale:
            h, w = img_meta[0]['img_shape'][:2]
            seg_logit = seg_logit[:, :, :h, :w]  # remove padding area
            seg_logit = torch.nn.functional.interpolate(seg_logit, size=tuple(img_meta[0]['ori_shape'][:2]),
                                                        mode='bilinear', align_corners=self.seg_head.align_corners)
        return seg_logit

    def inference_seg(self, img, img_meta, rescale):
        assert self.test_cfg['seg']['mode'] in ['whole']
        ori_shape = img_meta[0]['ori_shape']
        assert all(_['ori_shape'] == ori_shape for _ in img_meta)
        output = torch.softmax(self.whole_inference_seg(img, img_meta, rescale), dim=1)
        if img_meta[0].get('flip', False):
            direction = img_meta[0]['flip_direction']
            assert direction in ['horizontal', 'vertical']
            output = output.flip(dims=(3,)) if direction == 'horizontal' else output.flip(dims=(2,))
        return output

    def simple_test_seg(self, img, img_meta, rescale=True):
        seg_pred = self.inference_seg(img, img_meta, rescale).argmax(dim=1)
        return list(seg_pred.cpu().numpy())

    # -------------------------------------------------------------------------------------
    def load_state_dict(self, *args, **kwargs):
        ops.WPLANES.bump()  # parameters change in place: the pre-split weight planes are stale
        if ops.STATE.grad_sink is not None:
            ops.STATE.grad_sink.params_changed()  # ... and so are the value ranges the optimizer keeps for them
        return super().load_state_dict(*args, **kwargs)

    def forward(self, task, img, img_metas, return_loss=True, dataset_name=None, **kwargs):
        ops.WPLANES.begin(task)  # (the weight-plane sets this task uses are refreshed together)
        ops.RANGES.begin(img.device)  # (value-range slots of the GEMM operands: one generation per iteration)
        if return_loss:
            return self.forward_train(task=task, img=img, img_metas=img_metas, **kwargs)
        with torch.no_grad():
            return self.forward_test(task, img, img_metas, **kwargs)

    def train_step(self, data, optimizer=None):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        task = data.get('task', None)
        dataset_name = data.get('dataset_name', None)
        log_vars = add_prefix(log_vars, f'{task}.{dataset_name}')
        if hasattr(self, 'task_weight'):
            weight = self.task_weight[task]
            loss = loss * weight
            log_vars = log_vars.scaled(weight) if isinstance(log_vars, LazyLogVars) else \
                OrderedDict((k, v * weight) for k, v in log_vars.items())
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    def val_step(self, data, optimizer=None):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        log_vars = add_prefix(log_vars, f"{data.get('task', None)}.{data.get('dataset_name', None)}")
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    @staticmethod
    def pack_losses(losses):
        """-> (loss, names, packed): per-key means, `loss` = sum of the keys containing 'loss'
        (multitask_learner.py:274-287), and ONE device vector of all scalars (detached) so that a
        step needs a single device->host copy (and, distributed, a single all-reduce)."""
        vec = losses.pop('__packed__', None) if isinstance(losses, dict) else None
        if vec is not None:
            # a head that already holds its scalars as one vector in key order (DINOHead.loss_static): 3 launches
            # instead of one mean + add per key (and a select / expand / add chain per key in backward)
            names = list(losses.keys())
            assert vec.dim() == 1 and vec.numel() == len(names)
            if all('loss' in n for n in names):
                loss = vec.sum()
            else:
                mask = torch.tensor([1.0 if 'loss' in n else 0.0 for n in names], device=vec.device)
                loss = (vec * mask).sum()
            names.append('loss')
            return loss, names, torch.cat([vec.detach().float(), loss.detach().float().reshape(1)])
        names, vals = [], []
        for loss_name, loss_value in losses.items():
            if isinstance(loss_value, torch.Tensor):
                vals.append(loss_value.mean())
            elif isinstance(loss_value, list):
                vals.append(sum(_loss.mean() for _loss in loss_value))
            else:
                raise TypeError(f'{loss_name} is not a tensor or list of tensors')
            names.append(loss_name)
        loss = sum(v for n, v in zip(names, vals) if 'loss' in n)
        names.append('loss')
        vals.append(loss)
        packed = torch.stack([v.detach().float().reshape(()) for v in vals])
        return loss, names, packed

    def _parse_losses(self, losses):
        loss, names, packed = MTL.pack_losses(losses)
        if dist.is_available() and dist.is_initialized() and not getattr(self, 'defer_log_allreduce', False):
            world = dist.get_world_size()
            # rank-consistency guard of the reference (multitask_learner.py:289-296) rides along
            from .dist import mean_over_ranks
            packed = torch.cat([packed, packed.new_tensor([float(len(names))])]).float().contiguous()
            host = mean_over_ranks(packed).tolist()
            # (the mean of `world` equal counts is not exact in fp32 for a world size that is no power of two — 7 keys on 6 ranks
            #  come back as 6.9999995 — while one differing rank moves it by at least 1 / world: ADVICE r5)
            assert abs(host[-1] - len(names)) < 0.5 / world, \
                'loss log variables are different across GPUs!\n' + f'rank {dist.get_rank()} keys: ' + ','.join(names)
            host = host[:-1]
        else:
            return loss, LazyLogVars(names, packed)  # one device->host copy, deferred to the first read
        return loss, OrderedDict(zip(names, host))

    # -------------------------------------------------------------------------------------
    def load_task_pretrain(self):
        """multitask_learner.py:308-353: remap `bbox_head.transformer.encoder.*` ->
        `shared_encoder.*`, drop `neck.*conv.bias`, load non-strictly."""
        if self.task_pretrain is None:
            print('You did not set task_pretrain, hence it is skipped.')
            return None
        rule = self.task_pretrain.get('rule', None)
        sd = torch.load(self.task_pretrain['pretrained'], map_location='cpu', weights_only=True)
        if 'state_dict' in sd:
            sd = sd['state_dict']
        if rule == 'dino_mmdet':
            out = OrderedDict()
            for name, param in sd.items():
                if name.startswith('neck') and name.endswith('conv.bias'):
                    continue
                new = name.replace('bbox_head.transformer.encoder', 'shared_encoder', 1) \
                    if name.startswith('bbox_head.transformer.encoder') else name
                assert new not in out, f'{name}-->{new}'
                out[new] = param
            sd = out
        return self.load_state_dict(sd, strict=False)

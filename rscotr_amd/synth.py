"""Seeded synthetic batches for the three tasks: the benchmark and the parity
tests feed the product path and the oracle from this one generator.  Shapes follow what the
reference's pipelines hand to `MTL.forward` (configs/_base_/{cls,det,seg}/*.py), data are random.
"""
import numpy as np
import torch

TASK_DATASET = dict(cls='resisc', det='dior', seg='potsdam')


def make_batch(task, batch_size=2, size=512, seed=0, device='cpu', num_cls=45, num_det=20, num_seg=5,
               max_gt=20):
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    img = torch.randn(batch_size, 3, size, size, generator=g)
    metas = [dict(img_shape=(size, size, 3), ori_shape=(size, size, 3), pad_shape=(size, size, 3),
                  scale_factor=1.0, flip=False, filename=f'synthetic_{seed}_{i}') for i in range(batch_size)]
    batch = dict(task=task, dataset_name=TASK_DATASET[task], img=img.to(device), img_metas=metas)
    if task == 'cls':
        batch['gt_label'] = torch.from_numpy(rs.randint(0, num_cls, batch_size)).long().to(device)
    elif task == 'det':
        boxes, labels, hboxes, hlabels = [], [], [], []
        for _ in range(batch_size):
            G = int(rs.randint(1, max_gt + 1))
            cxy = rs.uniform(0.1, 0.9, (G, 2)) * size
            wh = rs.uniform(16, min(200, size / 2), (G, 2))
            b = np.concatenate([cxy - wh / 2, cxy + wh / 2], 1).clip(0, size).astype(np.float32)
            lab = rs.randint(0, num_det, G).astype(np.int64)
            boxes.append(torch.from_numpy(b).to(device))
            labels.append(torch.from_numpy(lab).to(device))
            hboxes.append(b)
            hlabels.append(lab)
        batch['gt_bboxes'], batch['gt_labels'] = boxes, labels
        # host copies of the ground truth, explicitly in the batch (the det head lays a batch out on the host when it gets
        # them: DetStatic checks them against the device tensors' shapes)
        batch['gt_bboxes_host'], batch['gt_labels_host'] = hboxes, hlabels
    elif task == 'seg':
        blk = 32 if size >= 64 else 8
        coarse = rs.randint(0, num_seg, (batch_size, 1, (size + blk - 1) // blk, (size + blk - 1) // blk))
        lab = np.kron(coarse, np.ones((1, 1, blk, blk), dtype=np.int64))[:, :, :size, :size]
        lab[rs.uniform(size=lab.shape) < 0.02] = 255
        batch['gt_semantic_seg'] = torch.from_numpy(lab).long().to(device)
    else:
        raise ValueError(task)
    return batch


def make_rnd(model, batch, seed=0, device='cpu', drop_path=True):
    """Explicit stochastic draws for one step: DropPath keep flags, the cls augment, CDN noise."""
    g = torch.Generator().manual_seed(seed + 12345)
    B = batch['img'].shape[0]
    rnd = {}
    rates = torch.tensor(model.backbone.drop_path_rates).repeat_interleave(2)
    if drop_path and float(rates.max()) > 0:
        rnd['drop_keep'] = torch.floor((1 - rates)[:, None] + torch.rand(rates.shape[0], B, generator=g)).to(device)
    else:
        rnd['drop_keep'] = None
    if batch['task'] == 'cls' and model.cls_augments is not None:
        rnd['cls_aug'] = model.cls_augments.draw(B, batch['img'].shape[-2:], np.random.RandomState(seed + 7))
    if batch['task'] == 'det':
        gen = model.bbox_head.dn_generator
        counts = [int(l.shape[0]) for l in batch['gt_labels']]
        ng = gen.get_num_groups(max(counts))
        K = 2 * ng * sum(counts)
        rnd['cdn'] = dict(label_p=torch.rand(K, generator=g).to(device),
                          new_label=torch.randint(0, gen.num_classes, (K,), generator=g).to(device),
                          rand_sign=torch.randint(0, 2, (K, 4), generator=g).float().to(device),
                          rand_part=torch.rand(K, 4, generator=g).to(device))
    return rnd


def to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_cpu(v) for v in obj)
    return obj

"""Distributed scalar helpers (packed: one all-reduce per call-site group, no host sync)."""
import torch

# ------------------------------------------------------------------------------------------
# distributed scalar helpers (packed: one all-reduce per call site group, no host sync)
# ------------------------------------------------------------------------------------------
def dist_world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def dist_mean_tensor(t):
    """reduce_mean of a small device vector (one all-reduce); identity in a single process."""
    if dist_world() == 1:
        return t
    import torch.distributed as dist
    t = t / dist.get_world_size()
    dist.all_reduce(t)
    return t


def dist_mean_vec(values, device):
    """mmdet reduce_mean for a list of host scalars in ONE all-reduce.  Single process: returns the
    python floats unchanged.  Distributed: returns 0-d device tensors (no host sync)."""
    if dist_world() == 1:
        return [float(v) for v in values]
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float32, device=device)
    dist.all_reduce(t.div_(dist.get_world_size()))
    return list(t.unbind(0))


# Host-side control group (gloo, created by the runner): small HOST vectors every rank must agree on or average — the det
# normalisers (functions of ground-truth counts, known on the host without a device sync) and the graph-or-eager flag of a det
# batch — travel through it in one all-reduce and never touch the device queue.
_HOST_GROUP = [None]


def set_host_group(group):
    _HOST_GROUP[0] = group


def host_group():
    return _HOST_GROUP[0]


def host_allreduce_sum(vec):
    """SUM over the ranks of a small float32 NumPy vector through the host control group (in place, returned)."""
    import numpy as np
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float32))
    dist.all_reduce(t, group=_HOST_GROUP[0])
    return t.numpy()


def clamp_min(x, lo):
    return x.clamp(min=lo) if torch.is_tensor(x) else max(x, lo)

"""Operator layer of the product path: torch.autograd.Function wrappers over the C ABI.

Each op launches hand-written gfx950 kernels from librscotr.so on the current torch stream with raw device pointers
(PyTorch only provides memory, streams and autograd bookkeeping).  There is NO CPU / eager fallback: a missing library, a
CPU tensor, or a non-zero return code raises.

    state      STATE: strategy switches and hooks (gradient sink, profiling, side stream) — the only mutable globals
    core       stream handle, workspace, profiling context, tensor checks
    ranges     RANGES: value ranges (amax slots) of GEMM operands for the fp16 split product
    matmul     gemm / gemm_batched, deferred split-K + grouped weight gradients (DEFER), weight planes (WPLANES), linear / mlp
    norm       layer_norm, layer_norm_fork, group_norm_tokens
    deform     msda, msda_prep, msda_attention (mmcv MultiScaleDeformableAttention)
    attention  swin_window_attention, mha (nn.MultiheadAttention), mask_logits, seg_attn_mask
    glue       level_embed_add, fan_out, cdn_queries, sine_embed4, neck im2col, layout helpers, cls pooling / loss
    losses     box utilities, match_cost_batched, lsap_*, focal / box loss sums, upsample_ce
    distutil   packed scalar all-reduces

Callers use `from rscotr_amd import ops; ops.linear(...)`: every public (and test-visible) name of the submodules is
re-exported here (submodule names differ from every op name: `ops.gemm` and `ops.msda` are the functions)."""
from . import attention, core, deform, distutil, glue, losses, matmul, norm, ranges, state
from .state import STATE

for _m in (core, ranges, matmul, norm, deform, attention, glue, losses, distutil):
    for _k, _v in vars(_m).items():
        if not _k.startswith('__') and _k not in ('STATE',):
            globals().setdefault(_k, _v)
del _m, _k, _v

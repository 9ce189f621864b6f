"""oracle — CPU restatement of the RSCoTr multi-task co-training step (TEST INFRASTRUCTURE).

This package is the checker, never the product:
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * nothing under `rscotr_amd/` imports it, and the product path has no CPU fallback.

What it restates
  Plain PyTorch (fp32, CPU) functions over a flat `params: dict[str, Tensor]` keyed by the
  reference's state-dict names.  Each function cites the reference
  file:line it follows.  Where the arithmetic lives in an un-vendored dependency of the
  reference (mmcv-full 1.6.1, mmdet 2.25.1, mmsegmentation 0.28.0, mmcls (unpinned), torch 1.11,
  scipy (unpinned) — README.md:71-82, requirement.txt:1-3 of the reference) the published
  algorithm of that pinned version is restated and the reference's own call site is cited.

PARITY UNPINNED by the reference, its self-contained functions excepted: the reference ships
no tests, golden vectors or fixtures for this path and cannot be imported
here (mmcv/mmdet/mmseg/mmcls absent, no network).  What of it runs without those libraries —
gen_sineembed_for_position, extract_dn_outputs, get_num_groups, Mask2FormerHead.forward_head,
MTL._parse_losses, the iteration strategies' __call__ — was run as it stands in the
build container and its outputs are committed (tests/golden/reference_static.npz,
tests/golden/make_reference_golden.py).  The rest of the restatement is pinned by independent primitives instead
(`F.grid_sample`, `torch.nn.MultiheadAttention`, `F.layer_norm/group_norm/conv2d/unfold`,
`torch.optim.AdamW`, `scipy.optimize.linear_sum_assignment`) and by closed-form known-answer
tests; see tests/test_oracle_*.py and DESIGN.md.
"""

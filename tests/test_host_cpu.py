"""Host-side pieces around the step that have no device work: the six iteration strategies and the MultiDataLoader of
mtl/data (iteration_strategies.py:66-258, multi_data_loader.py:109-191), the config loader on the reference's own config
file, and the Swin checkpoint converter."""
import os

import numpy as np
import pytest
import torch

from rscotr_amd import data as D


class _DS:
    def __init__(self, n, task):
        self.n, self.task = n, task

    def __len__(self):
        return self.n


class _Loader:
    def __init__(self, name, n, task, batches):
        self.dataset, self.name, self.batches = _DS(n, task), name, batches

    def __len__(self):
        return self.batches

    def __iter__(self):
        return iter([dict(src=self.name, k=i) for i in range(self.batches)])


def _loaders():
    return dict(resisc=_Loader('resisc', 300, 'cls', 3), dior=_Loader('dior', 100, 'det', 2), potsdam=_Loader('potsdam', 600, 'seg', 4))


def test_deterministic_strategies():
    L = _loaders()
    assert [D.RoundRobinIterationStrategy(L)() for _ in range(1)] == [0]
    rr = D.RoundRobinIterationStrategy(L, start_idx=1)
    assert [rr() for _ in range(7)] == [1, 2, 0, 1, 2, 0, 1]
    c = D.ConstantIterationStrategy(L, idx=2)
    assert [c() for _ in range(3)] == [2, 2, 2] and c.should_exhaust_all_iterators
    rs = D.RepeatedSequenceIterationStrategy(L, sequence=[0, 0, 1, 2])
    assert [rs() for _ in range(9)] == [0, 0, 1, 2, 0, 0, 1, 2, 0]
    with pytest.raises(AssertionError):
        D.RepeatedSequenceIterationStrategy(L, sequence=[0, 1])  # must name every loader


def test_random_strategies_draw_like_the_reference():
    """The reference draws np.random.choice(n, 1[, p])[0] from the GLOBAL NumPy stream on every call (same seed on
    all ranks => same task order): the sequences must be the ones that stream yields."""
    L = _loaders()
    np.random.seed(7)
    want = [np.random.choice(3, 1)[0] for _ in range(30)]
    np.random.seed(7)
    s = D.RandomIterationStrategy(L)
    assert [s() for _ in range(30)] == want
    p = [394 / 7984, 5862 / 7984, 1728 / 7984]  # configs/multi/slvl_strategies/batch-weighted_random.py
    np.random.seed(11)
    want = [np.random.choice(3, 1, p=p)[0] for _ in range(30)]
    np.random.seed(11)
    s = D.WeightedRandomIterationStrategy(L, p=[394, 5862, 1728])
    assert [s() for _ in range(30)] == want
    np.random.seed(3)
    want = [np.random.choice(3, 1, p=[0.3, 0.1, 0.6])[0] for _ in range(30)]
    np.random.seed(3)
    s = D.SizeProportionalIterationStrategy(L)
    assert [s() for _ in range(30)] == want and s.should_exhaust_all_iterators


def test_build_strategy_burns_300_probe_draws():
    """mtl/data/build.py:78-87: a second instance draws 300 indices at start-up (advancing the global stream)."""
    L = _loaders()
    np.random.seed(5)
    for _ in range(300):
        np.random.choice(3, 1)
    want = [np.random.choice(3, 1)[0] for _ in range(5)]
    np.random.seed(5)
    s = D.build_iteration_strategy(dict(strategy=dict(type='random')), L)
    assert [s() for _ in range(5)] == want
    assert isinstance(D.build_iteration_strategy({}, L), D.RoundRobinIterationStrategy)


def test_multi_data_loader_tags_and_restarts():
    L = _loaders()
    m = D.MultiDataLoader(L, D.RoundRobinIterationStrategy(L))
    assert len(m) == 9 and m.dataset_list == ['resisc', 'dior', 'potsdam']
    it = iter(m)
    got = [next(it) for _ in range(9)]
    assert [(b['dataset_name'], b['task']) for b in got[:3]] == [('resisc', 'cls'), ('dior', 'det'), ('potsdam', 'seg')]
    assert all(b['src'] == b['dataset_name'] for b in got)
    # dior has 2 batches: its third turn restarts the exhausted iterator (multi_data_loader.py:163-166)
    assert [b['k'] for b in got if b['src'] == 'dior'] == [0, 1, 0]
    # an exhausting strategy (should_exhaust_all_iterators) marks finished loaders, re-draws past them and stops
    # once every loader ran out: one epoch = every batch of every loader exactly once (multi_data_loader.py:157-162).
    # (A ConstantIterationStrategy would spin forever in change_dataloader here, in the reference as well: it can
    # only ever name its one loader.)
    class _Cycle(D.RoundRobinIterationStrategy):
        should_exhaust_all_iterators = True

    m = D.MultiDataLoader(L, _Cycle(L))
    got = [b for b in m]
    assert len(got) == 9
    assert sorted((b['src'], b['k']) for b in got) == sorted((n, i) for n, l in L.items() for i in range(l.batches))


def _check_main_cfg(cfg):
    assert cfg.model['type'] == 'MTL' and cfg.model['backbone']['type'] == 'SwinTransformer'
    assert cfg.model['bbox_head']['num_query'] == 600 and cfg.model['task_weight']['seg'] == 0.1
    assert cfg.optimizer['type'] == 'AdamW' and cfg.optimizer_config['grad_clip']['max_norm'] == 0.1
    assert cfg.dist_params['backend'] == 'nccl'  # inherited from default_runtime.py through _base_


def test_repo_config_loads():
    """The repo's own (rewritten, helper-based) main config."""
    from rscotr_amd import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _check_main_cfg(Config.fromfile(os.path.join(root, 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')))


REF_MULTI = '/root/reference/configs/multi'


@pytest.mark.skipif(not os.path.isdir(REF_MULTI), reason='the reference tree only exists in the build container')
def test_reference_configs_load_unchanged_and_build():
    """Drop-in boundary, config level: the REFERENCE's own config files (read where they lie, never copied) load with
    rscotr_amd.Config, equal the repo's rewritten ones in every model / optimizer / schedule field, and MODELS.build
    turns the main one into an MTL with the reference's parameter count."""
    import glob
    from rscotr_amd import Config, MODELS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(REF_MULTI, '*.py')) + glob.glob(os.path.join(REF_MULTI, 'slvl_strategies', '*.py')))
    assert len(files) >= 8
    for path in files:
        if os.path.basename(path) == 'default_runtime.py':
            continue
        ref = Config.fromfile(path)
        assert ref.model['type'] == 'MTL', path
        mine = os.path.join(root, 'configs', 'multi', os.path.relpath(path, REF_MULTI))
        if os.path.exists(mine):
            own = Config.fromfile(mine)
            for key in ('model', 'optimizer', 'optimizer_config', 'lr_config', 'runner'):
                assert _plain(ref.get(key)) == _plain(own.get(key)), (os.path.basename(path), key)
    ref = Config.fromfile(os.path.join(REF_MULTI, 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py'))
    _check_main_cfg(ref)
    import copy
    model = MODELS.build(copy.deepcopy(ref.model))
    n = sum(p.numel() for p in model.parameters())
    assert abs(n - 62.55e6) < 0.05e6, n


def _plain(o):
    """dict / list skeleton of a config value; machine-local checkpoint paths (the reference hard-codes /home/rs/...,
    the repo's configs leave them None) compare equal."""
    if isinstance(o, str) and o.startswith('/home/'):
        return None
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    return o


def test_swin_converter_permutes_patch_merging():
    """Official PatchMerging ([x00, x10, x01, x11] concat -> LN -> Linear) and the Unfold-ordered one of this repo with
    the converted weights compute the same thing; key names land in the mmdet layout."""
    from rscotr_amd import ops
    from rscotr_amd.checkpoint import swin_converter
    g = torch.Generator().manual_seed(0)
    C, H, W, B = 8, 6, 4, 2
    x = torch.randn(B, H * W, C, generator=g)
    red = torch.randn(2 * C, 4 * C, generator=g)
    nw, nb = torch.randn(4 * C, generator=g), torch.randn(4 * C, generator=g)
    xs = x.view(B, H, W, C)
    off = torch.cat([xs[:, 0::2, 0::2], xs[:, 1::2, 0::2], xs[:, 0::2, 1::2], xs[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    want = torch.nn.functional.linear(torch.nn.functional.layer_norm(off, (4 * C,), nw, nb), red)
    sd = swin_converter({'layers.0.downsample.reduction.weight': red, 'layers.0.downsample.norm.weight': nw,
                         'layers.0.downsample.norm.bias': nb, 'layers.1.blocks.0.attn.qkv.weight': torch.zeros(3, 3),
                         'layers.1.blocks.0.mlp.fc1.bias': torch.zeros(3), 'patch_embed.proj.weight': torch.zeros(1),
                         'head.weight': torch.zeros(1), 'norm.weight': torch.zeros(1)})
    assert set(sd) == {'backbone.stages.0.downsample.reduction.weight', 'backbone.stages.0.downsample.norm.weight',
                       'backbone.stages.0.downsample.norm.bias', 'backbone.stages.1.blocks.0.attn.w_msa.qkv.weight',
                       'backbone.stages.1.blocks.0.ffn.layers.0.0.bias', 'backbone.patch_embed.projection.weight',
                       'backbone.norm.weight'}
    y, _ = ops.patch_merge_gather(x, (H, W))  # Unfold order c*4 + kh*2 + kw (pure indexing: runs on the CPU)
    got = torch.nn.functional.linear(
        torch.nn.functional.layer_norm(y, (4 * C,), sd['backbone.stages.0.downsample.norm.weight'],
                                       sd['backbone.stages.0.downsample.norm.bias']),
        sd['backbone.stages.0.downsample.reduction.weight'])
    assert torch.allclose(got, want, atol=1e-5)


# ---- checkpoint interop ------------------------------------------------------------------
A8_PATTERNS = [  # the reference's state-dict key names (attribute names of the mm* modules it builds)
    r'backbone\.patch_embed\.(projection|norm)\.(weight|bias)',
    r'backbone\.stages\.\d\.blocks\.\d+\.(norm1|norm2)\.(weight|bias)',
    r'backbone\.stages\.\d\.blocks\.\d+\.attn\.w_msa\.(relative_position_bias_table|relative_position_index)',
    r'backbone\.stages\.\d\.blocks\.\d+\.attn\.w_msa\.(qkv|proj)\.(weight|bias)',
    r'backbone\.stages\.\d\.blocks\.\d+\.ffn\.layers\.(0\.0|1)\.(weight|bias)',
    r'backbone\.stages\.[012]\.downsample\.(norm\.(weight|bias)|reduction\.weight)',
    r'backbone\.norm[0-3]\.(weight|bias)',
    r'neck\.(convs\.[012]|extra_convs\.0)\.(conv\.weight|gn\.(weight|bias))',
    r'shared_encoder\.layers\.[0-5]\.attentions\.0\.(sampling_offsets|attention_weights|value_proj|output_proj)\.(weight|bias)',
    r'shared_encoder\.layers\.[0-5]\.ffns\.0\.layers\.(0\.0|1)\.(weight|bias)',
    r'shared_encoder\.layers\.[0-5]\.norms\.[01]\.(weight|bias)',
    r'cls_head\.fc\.(weight|bias)',
    r'bbox_head\.cls_branches\.[0-6]\.(weight|bias)',
    r'bbox_head\.reg_branches\.[0-6]\.[024]\.(weight|bias)',
    r'bbox_head\.label_embedding\.weight',
    r'bbox_head\.transformer\.(level_embeds|enc_output\.(weight|bias)|enc_output_norm\.(weight|bias)|query_embed\.weight)',
    r'bbox_head\.transformer\.decoder\.layers\.[0-5]\.attentions\.0\.attn\.(in_proj_weight|in_proj_bias|out_proj\.(weight|bias))',
    r'bbox_head\.transformer\.decoder\.layers\.[0-5]\.attentions\.1\.(sampling_offsets|attention_weights|value_proj|output_proj)\.(weight|bias)',
    r'bbox_head\.transformer\.decoder\.layers\.[0-5]\.ffns\.0\.layers\.(0\.0|1)\.(weight|bias)',
    r'bbox_head\.transformer\.decoder\.layers\.[0-5]\.norms\.[012]\.(weight|bias)',
    r'bbox_head\.transformer\.decoder\.(ref_point_head\.[02]|norm)\.(weight|bias)',
    r'seg_head\.pixel_decoder\.(level_encoding\.weight|mask_feature\.(weight|bias))',
    r'seg_head\.transformer_decoder\.layers\.[0-8]\.attentions\.[01]\.attn\.(in_proj_weight|in_proj_bias|out_proj\.(weight|bias))',
    r'seg_head\.transformer_decoder\.layers\.[0-8]\.ffns\.0\.layers\.(0\.0|1)\.(weight|bias)',
    r'seg_head\.transformer_decoder\.layers\.[0-8]\.norms\.[012]\.(weight|bias)',
    r'seg_head\.transformer_decoder\.post_norm\.(weight|bias)',
    r'seg_head\.(query_embed|query_feat|level_embed)\.weight',
    r'seg_head\.mask_embed\.[024]\.(weight|bias)',
]


@pytest.fixture(scope='module')
def main_model():
    from util import build_model, load_model_cfg
    cfg, mcfg = load_model_cfg(tiny=False)
    return cfg, mcfg, build_model(mcfg, perturb=False)


def test_state_dict_keys_follow_the_reference_names(main_model):
    """Every key of the built model is one of the reference's names and every name family occurs: a reference `.pth`
    (HF Qingyun/RSCoTr) loads by name, and a checkpoint written here loads in the reference."""
    import re
    cfg, mcfg, model = main_model
    pats = [re.compile(p + '$') for p in A8_PATTERNS]
    hits = [0] * len(pats)
    for k in model.state_dict():
        m = [i for i, p in enumerate(pats) if p.match(k)]
        assert m, f'state-dict key outside the reference naming: {k}'
        hits[m[0]] += 1
    assert all(hits), [A8_PATTERNS[i] for i, h in enumerate(hits) if not h]
    sd = model.state_dict()
    assert sd['backbone.stages.0.downsample.reduction.weight'].shape == (192, 384)
    assert sd['bbox_head.transformer.decoder.layers.0.attentions.0.attn.in_proj_weight'].shape == (768, 256)
    assert sd['backbone.stages.0.blocks.0.attn.w_msa.relative_position_bias_table'].shape == (169, 3)


def test_task_pretrain_remap(main_model, tmp_path):
    """multitask_learner.py:308-353, rule 'dino_mmdet': an mmdet DINO checkpoint keeps its encoder under
    bbox_head.transformer.encoder.* and has biased neck convs; the encoder lands in shared_encoder.*, the conv biases
    are dropped, everything else loads by name (non-strict)."""
    cfg, mcfg, model = main_model
    g = torch.Generator().manual_seed(0)
    src = {}
    for k, v in model.state_dict().items():
        if k.startswith(('cls_head', 'seg_head')):
            continue  # a det checkpoint has neither
        nk = k.replace('shared_encoder.', 'bbox_head.transformer.encoder.', 1) if k.startswith('shared_encoder.') else k
        src[nk] = torch.randn(v.shape, generator=g) if v.dtype.is_floating_point else v.clone()
    for i in range(3):
        src[f'neck.convs.{i}.conv.bias'] = torch.randn(256, generator=g)
    path = str(tmp_path / 'dino.pth')
    torch.save(dict(state_dict=src, meta=dict(iter=7)), path)
    old = model.task_pretrain
    before = {k: v.clone() for k, v in model.state_dict().items()}
    try:
        model.task_pretrain = dict(rule='dino_mmdet', pretrained=path)
        report = model.load_task_pretrain()
        sd = model.state_dict()
        assert not report.unexpected_keys
        assert all(k.startswith(('cls_head', 'seg_head')) for k in report.missing_keys) and report.missing_keys
        k = 'shared_encoder.layers.3.ffns.0.layers.0.0.weight'
        assert torch.equal(sd[k], src[k.replace('shared_encoder.', 'bbox_head.transformer.encoder.', 1)])
        assert torch.equal(sd['bbox_head.cls_branches.6.weight'], src['bbox_head.cls_branches.6.weight'])
        assert torch.equal(sd['seg_head.query_feat.weight'], before['seg_head.query_feat.weight'])
    finally:
        model.task_pretrain = old
        model.load_state_dict(before)


def test_checkpoint_round_trip_and_resume(tmp_path):
    """save_checkpoint writes the mmcv layout (meta / state_dict / optimizer in torch.optim.AdamW's state layout);
    resume restores weights, Adam moments, per-parameter step counts and the iteration counter into a fresh run, also
    from a checkpoint saved under a DDP wrapper (`module.` prefix)."""
    from util import build_model, load_model_cfg
    from rscotr_amd.checkpoint import load_checkpoint, resume, save_checkpoint
    from rscotr_amd.optim import build_optimizer
    cfg, mcfg = load_model_cfg(tiny=True)
    model = build_model(mcfg, seed=1)
    opt = build_optimizer(model, cfg['optimizer'], cfg.get('optimizer_config'))
    g = torch.Generator().manual_seed(3)
    live = [i for i, gr in enumerate(opt.groups) if gr['name'].startswith(('backbone', 'cls_head'))]
    for i in live:  # as if a few cls steps had run
        o, n = opt.offsets[i], opt.groups[i]['param'].numel()
        opt.flat_m[o:o + n] = torch.randn(n, generator=g)
        opt.flat_v[o:o + n] = torch.rand(n, generator=g)
        opt.steps[i], opt.live[i] = 5, True
    model.CLASSES = dict(resisc=('a', 'b'))
    path = str(tmp_path / 'iter_5.pth')
    ck = save_checkpoint(path, model, opt, meta=dict(iter=5))
    assert set(ck) == {'meta', 'state_dict', 'optimizer'} and set(ck['optimizer']) == {'state', 'param_groups'}
    assert len(ck['optimizer']['param_groups']) == len(opt.groups) and sorted(ck['optimizer']['state']) == live
    # the optimizer entry is what torch.optim.AdamW itself would load
    ref_opt = torch.optim.AdamW([dict(params=[gr['param']], lr=gr['lr'], weight_decay=gr['weight_decay']) for gr in opt.groups])
    ref_opt.load_state_dict(torch.load(path, weights_only=True)['optimizer'])
    assert int(ref_opt.state[opt.groups[live[0]]['param']]['step']) == 5

    class _R:  # the two attributes resume() touches besides the model
        pass

    model2 = build_model(mcfg, seed=2)
    r = _R()
    r.model, r.optimizer, r.iter = model2, build_optimizer(model2, cfg['optimizer'], cfg.get('optimizer_config')), 0
    resume(r, path)
    assert r.iter == 5 and model2.CLASSES == dict(resisc=('a', 'b'))
    for (k, a), b in zip(model.state_dict().items(), model2.state_dict().values()):
        assert torch.equal(a, b), k
    assert torch.equal(r.optimizer.flat_m, opt.flat_m) and torch.equal(r.optimizer.flat_v, opt.flat_v)
    assert (r.optimizer.steps == opt.steps).all() and (r.optimizer.live == opt.live).all()
    # parameters stayed views of the arena (load copies in place)
    p0 = r.optimizer.groups[0]['param']
    assert p0.data_ptr() == r.optimizer.flat_p[r.optimizer.offsets[0]:].data_ptr()
    # DDP-saved checkpoint
    torch.save(dict(state_dict={'module.' + k: v for k, v in ck['state_dict'].items()}), str(tmp_path / 'ddp.pth'))
    model3 = build_model(mcfg, seed=4)
    _, rep = load_checkpoint(model3, str(tmp_path / 'ddp.pth'), strict=True)
    assert not rep.missing_keys and not rep.unexpected_keys


def test_c_abi_library_exports_every_declared_symbol():
    """include/rscotr.h is the boundary: the built library must export exactly the entry points it declares, with no
    torch or C++ types in any signature (no compute call here: this runs without a GPU)."""
    import ctypes
    from rscotr_amd import _lib
    sigs = _lib.parse_header()
    assert len(sigs) >= 40 and 'rscotr_msda_fwd' in sigs and 'rscotr_gemm_f32' in sigs and 'rscotr_lsap_dev_f32' in sigs
    dll = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sigs if not hasattr(dll, n)]
    assert not missing, missing
    import re
    src = open(_lib.HEADER).read()
    code = re.sub(r'//[^\n]*', '', re.sub(r'/\*.*?\*/', '', src, flags=re.S))  # declarations without the comments
    assert 'extern "C"' in code and not re.search(r'torch|Tensor|at::|std::|template', code)
    dll.rscotr_last_error.restype = ctypes.c_char_p
    dll.rscotr_version.restype = ctypes.c_char_p if sigs['rscotr_version'][0] is ctypes.c_char_p else ctypes.c_int
    assert dll.rscotr_version() is not None
    # the shared object says which revision of the argument lists it was built with; the binding refuses any other
    assert dll.rscotr_version() == _lib.header_abi_version() >= 7
    # host-side argument checking works without a device: a negative dimension is refused with a message
    dll.rscotr_gemm_f32_workspace.restype = ctypes.c_int64
    assert dll.rscotr_gemm_f32_workspace(-1, 4, 4) == 0
    assert dll.rscotr_gemm_f32_workspace(256, 256, 10880) > 0


def test_state_dict_key_families_the_reference_defines_itself(main_model):
    """The part of SURVEY A.8 that needs no memory of mmcv / mmdet: the attribute names under which the reference's OWN classes
    register their parameter-carrying members (`self.<name> = <constructor>(...)`, collected with `ast` by
    tests/golden/make_reference_golden.py).  Every such member that this configuration builds must own state-dict keys under
    exactly that name at the class's place in the model."""
    import numpy as np
    cfg, mcfg, model = main_model
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_static.npz'))
    fam = [r.split(':') for r in z['attr_families'].tolist()]
    place = {'MTL': '', 'DINOHead': 'bbox_head.', 'DeformableDETRHead': 'bbox_head.', 'DETRHead': 'bbox_head.',
             'DinoTransformer': 'bbox_head.transformer.', 'DinoTransformerDecoder': 'bbox_head.transformer.decoder.',
             'Mask2FormerHead': 'seg_head.', 'MlvlSegPixelDecoder': 'seg_head.pixel_decoder.', 'SlvlClsHead': 'cls_head.'}
    carriers = ('nn.Embedding', 'nn.Parameter', 'nn.Linear', 'nn.LayerNorm', 'nn.Sequential', 'Conv2d', 'Linear', 'build_MLP',
                'build_transformer_layer_sequence', 'build_transformer', 'build_backbone', 'build_neck', 'build_head', '_get_clones')
    # members the reference builds only under another configuration: one-stage queries (deformable_detr_head.py:78-80), the
    # class branch of scheme 1 (mask2former_head.py:78-79), DETRHead's own layers (its _init_layers is overridden)
    not_built = {('DeformableDETRHead', 'query_embedding'), ('DETRHead', 'query_embedding'), ('Mask2FormerHead', 'cls_embed'),
                 ('DETRHead', 'input_proj'), ('DETRHead', 'fc_cls'), ('DETRHead', 'fc_reg'), ('DETRHead', 'reg_ffn')}
    keys = list(model.state_dict())
    checked = 0
    for cls, attr, ctor in fam:
        if cls not in place or ctor not in carriers or (cls, attr) in not_built:
            continue
        prefix = place[cls] + attr
        assert any(k == prefix or k.startswith(prefix + '.') for k in keys), f'{cls}.{attr} ({ctor}): no state-dict key under {prefix!r}'
        checked += 1
    assert checked >= 20, checked
    # and nothing at those places that the reference's classes do not name
    named = {place[c] + a for c, a, _ in fam if c in place}
    for k in keys:
        for cls_place in ('bbox_head.transformer.decoder.', 'bbox_head.transformer.', 'seg_head.pixel_decoder.', 'seg_head.', 'bbox_head.', 'cls_head.'):
            if k.startswith(cls_place):
                member = cls_place + k[len(cls_place):].split('.')[0]
                if cls_place == 'bbox_head.transformer.decoder.' and member.endswith('.layers'):
                    break  # (TransformerLayerSequence.layers: mmcv's name)
                if cls_place == 'cls_head.' and member == 'cls_head.fc':
                    break  # (mmcls LinearClsHead.fc)
                assert member in named, f'{k}: {member} is not an attribute the reference classes assign'
                break


def test_reference_import_paths_and_train_model_signature(tmp_path):
    """VERDICT r4 missing 7: the reference's import paths (`mtl.apis`, `mtl.data`, `mtl.runner.hooks`, `mtl.utils.optimizer`,
    `models.multi`) resolve to this package, `train_model` has the signature of mtl/apis/train.py:24-30, and a config's
    `custom_imports` is honoured — an unknown module raises (unless the config allows failures), it is never ignored."""
    import inspect
    import rscotr_amd
    from rscotr_amd import Config
    from rscotr_amd.compat import apply_custom_imports, install_aliases
    assert len(install_aliases()) >= 18
    from mtl.apis import train_model
    from mtl.data import MultiDataLoader, RoundRobinIterationStrategy
    from mtl.engine import multi_gpu_test, single_gpu_test
    from mtl.runner.hooks import MultiDatasetsEvalHook
    from mtl.utils.optimizer import build_optimizer
    import models.multi
    assert MultiDataLoader is rscotr_amd.data.MultiDataLoader and models.multi.MTL is rscotr_amd.mtl.MTL
    assert callable(single_gpu_test) and callable(multi_gpu_test) and callable(build_optimizer)
    assert RoundRobinIterationStrategy is rscotr_amd.data.RoundRobinIterationStrategy
    assert MultiDatasetsEvalHook is rscotr_amd.engine.MultiDatasetsEvalHook
    want = ['model', 'datasets', 'cfg', 'distributed', 'validate', 'timestamp', 'meta']
    ref = '/root/reference/mtl/apis/train.py'
    if os.path.exists(ref):  # (build container only: the GPU box has no reference tree)
        import ast
        fn = [n for n in ast.parse(open(ref).read()).body if isinstance(n, ast.FunctionDef) and n.name == 'train_model'][0]
        want = [a.arg for a in fn.args.args]
        assert [ast.literal_eval(d) for d in fn.args.defaults] == [False, False, None, None]
    sig = inspect.signature(train_model)
    assert list(sig.parameters) == want
    assert [p.default for p in sig.parameters.values()][3:] == [False, False, None, None]
    # custom_imports: the reference's own value works; an unknown module is an error, or a warning when the config says so
    p = tmp_path / 'c.py'
    p.write_text("custom_imports = dict(imports='models.multi', allow_failed_imports=False)\nx = 1\n")
    assert Config.fromfile(str(p)).x == 1
    p.write_text("custom_imports = dict(imports=['models.multi', 'no_such_module_xyz'], allow_failed_imports=False)\n")
    with pytest.raises(ImportError):
        Config.fromfile(str(p))
    p.write_text("custom_imports = dict(imports=['no_such_module_xyz'], allow_failed_imports=True)\n")
    with pytest.warns(UserWarning):
        assert apply_custom_imports(Config.fromfile(str(p), import_custom_modules=False)) == []


def test_binding_refuses_a_library_of_another_abi_revision(monkeypatch, tmp_path):
    """A stale .so (or an A/B build picked by RSCOTR_LIB) whose entries have older argument lists must not be called."""
    from rscotr_amd import _lib
    hdr = tmp_path / 'rscotr.h'
    hdr.write_text(open(_lib.HEADER).read().replace(f'#define RSCOTR_ABI_VERSION {_lib.header_abi_version()}',
                                                     '#define RSCOTR_ABI_VERSION 9999'))
    monkeypatch.setattr(_lib, 'HEADER', str(hdr))
    real = _lib.header_abi_version
    monkeypatch.setattr(_lib, 'header_abi_version', lambda path=str(hdr): real(path))
    fresh = _lib._Lib()
    with pytest.raises(RuntimeError, match='ABI revision'):
        fresh.load()


def test_rank_consistency_guard_tolerates_an_inexact_mean():
    """The key count rides in the averaged loss vector: for a world size that is no power of two the mean of equal counts is
    not the count in fp32 (7 keys on 6 ranks: 6.9999995).  The guard must pass there and still fail when one rank differs."""
    import numpy as np
    for world in (2, 3, 5, 6, 7, 8):
        for n in (7, 13, 14, 25):
            mean = np.float32(0)
            for _ in range(world):
                mean = np.float32(mean + np.float32(n) / np.float32(world))
            assert abs(float(mean) - n) < 0.5 / world
            other = np.float32(mean + np.float32(1) / np.float32(world))  # one rank logs one key more
            assert not abs(float(other) - n) < 0.5 / world

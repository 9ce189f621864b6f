"""The reference's import paths and its training entry point, for callers that were written against them.

`install_aliases()` registers module objects under the names the reference's tools and configs import — `mtl.apis`,
`mtl.data`, `mtl.engine`, `mtl.model`, `mtl.runner.hooks`, `mtl.utils.optimizer`, `models.multi` (`tools/train.py:20-32`,
`configs/multi/*.py: custom_imports = dict(imports='models.multi')`) — each holding the names the reference's module of that
path exports, bound to this package's implementations.  Nothing is copied: the alias modules are created in `sys.modules` at
run time and only re-export.  `train_model` has the reference's signature (`mtl/apis/train.py:24-30`).
`apply_custom_imports(cfg)` is mmcv's `import_modules_from_strings` for a config's `custom_imports` (`tools/train.py:123-126`)."""
import importlib
import sys
import types
import warnings


def _module(name, **names):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__doc__ = f'alias of rscotr_amd for the reference import path `{name}` (rscotr_amd.compat)'
        m.__rscotr_alias__ = True
        sys.modules[name] = m
        parent, _, child = name.rpartition('.')
        if parent:
            setattr(_module(parent), child, m)
    for k, v in names.items():
        setattr(m, k, v)
    return m


def train_model(model, datasets, cfg, distributed=False, validate=False, timestamp=None, meta=None):
    """mtl/apis/train.py:24-118.  `datasets`: what the reference passes is a list of mm* dataset objects it builds loaders
    from; those libraries are not part of this package, so `datasets` is here one of: a `MultiDataLoader`; a
    {dataset name: loader} dict (wrapped with `cfg.data.iteration_strategy`, as `build_multidataloader` does); None = the
    synthetic device-side loaders of the config's shapes.  `distributed` must agree with the process group (the exchange
    is set up from `torch.distributed`, one process per GPU); `validate` needs `cfg.data.val_loaders` ({dataset: loader})."""
    import torch.distributed as dist
    from .data import MultiDataLoader, build_iteration_strategy, build_synthetic_multidataloader
    from .runner import build_runner
    if distributed != (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        raise ValueError(f'train_model(distributed={distributed}) does not match the process group '
                         '(launch one process per GPU with torch.distributed.run and call init_process_group first)')
    get = cfg.get if hasattr(cfg, 'get') else (lambda k, d=None: getattr(cfg, k, d))
    if isinstance(datasets, (list, tuple)) and len(datasets) == 1:
        datasets = datasets[0]
    if datasets is None:
        dev = next(model.parameters()).device
        loader = build_synthetic_multidataloader(cfg, dev, rank=dist.get_rank() if distributed else 0)
    elif isinstance(datasets, MultiDataLoader):
        loader = datasets
    elif isinstance(datasets, dict):
        scfg = (get('data') or {}).get('iteration_strategy', dict(type='RoundRobinIterationStrategy'))
        loader = MultiDataLoader(datasets, build_iteration_strategy(scfg, datasets))
    else:
        raise TypeError('train_model: datasets must be a MultiDataLoader, a {dataset name: loader} dict or None (synthetic): the '
                        'mm* dataset classes the reference builds its loaders from are outside this package')
    val = (get('data') or {}).get('val_loaders') if validate else None
    runner = build_runner(model, cfg, loader, val_dataloaders=val, validate=validate, meta=meta, timestamp=timestamp)
    runner.run()
    return runner


def install_aliases():
    """Idempotent; -> the names registered."""
    from . import data, engine, hooks, mtl, optim, registry, runner
    _module('mtl')
    _module('mtl.apis', train_model=train_model)
    _module('mtl.apis.train', train_model=train_model)
    strategies = {k: getattr(data, k) for k in dir(data) if k.endswith('IterationStrategy')}
    _module('mtl.data', MultiDataLoader=data.MultiDataLoader, build_iteration_strategy=data.build_iteration_strategy, **strategies)
    _module('mtl.data.multi_data_loader', MultiDataLoader=data.MultiDataLoader)
    _module('mtl.data.iteration_strategies', build_iteration_strategy=data.build_iteration_strategy, **strategies)
    _module('mtl.engine', single_gpu_test=engine.single_gpu_test, multi_gpu_test=engine.multi_gpu_test)
    _module('mtl.engine.test', single_gpu_test=engine.single_gpu_test, multi_gpu_test=engine.multi_gpu_test)
    builders = {k: getattr(registry, k) for k in dir(registry) if k.startswith('build_')}
    _module('mtl.model', MODELS=registry.MODELS, **builders)
    _module('mtl.model.build', MODELS=registry.MODELS, **builders)
    _module('mtl.runner', IterBasedRunner=runner.IterBasedRunner, build_runner=runner.build_runner)
    _module('mtl.runner.hooks', MultiDatasetsEvalHook=engine.MultiDatasetsEvalHook, CheckpointHook=hooks.CheckpointHook,
            TextLoggerHook=hooks.TextLoggerHook, TensorboardLoggerHook=hooks.TensorboardLoggerHook)
    _module('mtl.runner.hooks.evaluation', MultiDatasetsEvalHook=engine.MultiDatasetsEvalHook)
    _module('mtl.utils', build_optimizer=optim.build_optimizer)
    _module('mtl.utils.optimizer', build_optimizer=optim.build_optimizer, build_param_groups=optim.build_param_groups)
    _module('models')
    _module('models.multi', MTL=mtl.MTL, MODELS=registry.MODELS)
    _module('models.multi.multitask_learner', MTL=mtl.MTL)
    return sorted(n for n, m in sys.modules.items() if getattr(m, '__rscotr_alias__', False))


def apply_custom_imports(cfg):
    """`custom_imports = dict(imports=[...] | '...', allow_failed_imports=False)` of a config: the reference's own module
    paths resolve to the aliases above (importing them registers every `type=` name, which importing rscotr_amd has
    already done); any other module is imported as mmcv would; a failure raises unless the config allows it — it is never
    silently ignored."""
    ci = cfg.get('custom_imports') if hasattr(cfg, 'get') else None
    if not ci:
        return []
    names = ci.get('imports', [])
    names = [names] if isinstance(names, str) else list(names)
    allow = bool(ci.get('allow_failed_imports', False))
    install_aliases()
    done = []
    for n in names:
        try:
            done.append(importlib.import_module(n))
        except ImportError as e:
            if not allow:
                raise ImportError(f'custom_imports: cannot import {n!r} ({e}); rscotr_amd provides the reference paths '
                                  f'{[a for a in sys.modules if getattr(sys.modules[a], "__rscotr_alias__", False)][:6]} ... only') from e
            warnings.warn(f'custom_imports: {n!r} failed to import and was ignored (allow_failed_imports=True)')
    return done

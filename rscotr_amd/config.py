"""mmcv-style config loading for the reference's `configs/multi/*.py` files.

The reference drives everything from mmcv `Config.fromfile` python configs
(tools/train.py:119; configs/multi/MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py).
mmcv is not available on the target image, so this module re-creates the subset of its
behaviour those files rely on: attribute-access dicts, `_base_` inheritance (str or list,
relative to the file), `_delete_=True` replacement, `{{_base_.name}}` substitution and
`--cfg-options a.b=c` merging (tools/train.py:81-90,127-128).
"""
import ast
import copy
import os
import re

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'


class ConfigDict(dict):
    """dict with attribute access (mmcv ConfigDict / addict.Dict semantics for reads)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = _wrap(value)

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def copy(self):
        return ConfigDict(dict.copy(self))


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigDict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


def _merge(a, b):
    """mmcv Config._merge_a_into_b(a, b): a overrides b, recursively; `_delete_` replaces."""
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and isinstance(b[k], dict) and not v.get(DELETE_KEY, False):
            b[k] = _merge(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = copy.deepcopy(v)
    return b


_SUBST = re.compile(r'\{\{\s*_base_\.([\w\.]+)\s*\}\}')


def _load_file(path):
    path = os.path.abspath(path)
    with open(path, encoding='utf-8') as fh:
        text = fh.read()
    # `{{_base_.x}}` placeholders are replaced by a marker string first, then resolved
    # against the merged base (mmcv Config._pre_substitute_base_vars / _substitute_base_vars)
    placeholders = {}

    def _mark(m):
        key = f'_BASEVAR_{len(placeholders)}_'
        placeholders[key] = m.group(1)
        return f'"{key}"'

    text = _SUBST.sub(_mark, text)
    ns = {}
    exec(compile(ast.parse(text, filename=path), path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith('__') and not isinstance(v, type(os)) and not callable(v)}
    bases = cfg.pop(BASE_KEY, None)
    merged_base = {}
    if bases is not None:
        if isinstance(bases, str):
            bases = [bases]
        for b in bases:
            sub = _load_file(os.path.join(os.path.dirname(path), b))
            dup = set(sub) & set(merged_base)
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases: {sorted(dup)}')
            merged_base.update(sub)
    if placeholders:
        cfg = _resolve(cfg, placeholders, merged_base)
    return _merge(cfg, merged_base)


def _lookup(base, dotted):
    cur = base
    for part in dotted.split('.'):
        cur = cur[part]
    return copy.deepcopy(cur)


def _resolve(v, ph, base):
    if isinstance(v, dict):
        return {k: _resolve(x, ph, base) for k, x in v.items()}
    if isinstance(v, list):
        return [_resolve(x, ph, base) for x in v]
    if isinstance(v, tuple):
        return tuple(_resolve(x, ph, base) for x in v)
    if isinstance(v, str):
        if v in ph:
            return _lookup(base, ph[v])
        for key, dotted in ph.items():
            if key in v:
                v = v.replace(key, str(_lookup(base, dotted)))
    return v


class Config:
    """Minimal stand-in for mmcv.Config: `Config.fromfile(path)`, attribute + item access,
    `.get`, `.merge_from_dict` (for --cfg-options)."""

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def fromfile(path, import_custom_modules=True):
        cfg = Config(_load_file(path), filename=path)
        if import_custom_modules and cfg.get('custom_imports'):  # mmcv Config.fromfile(import_custom_modules=True)
            from .compat import apply_custom_imports
            apply_custom_imports(cfg)
        return cfg

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setitem__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)

    def keys(self):
        return self._cfg_dict.keys()

    def copy(self):
        return Config(copy.deepcopy(dict(self._cfg_dict)), self.filename)

    def merge_from_dict(self, options):
        """`--cfg-options model.backbone.depths=[2,2,2,2]` style dotted overrides."""
        nested = {}
        for full_key, v in options.items():
            d = nested
            parts = full_key.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        object.__setattr__(self, '_cfg_dict', _wrap(_merge(nested, self._cfg_dict)))

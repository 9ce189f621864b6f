"""Task alternation: iteration strategies + MultiDataLoader (mtl/data/iteration_strategies.py,
mtl/data/multi_data_loader.py, mtl/data/build.py:58-100), and seeded synthetic per-task loaders
that stand in for the RESISC45 / DIOR / Potsdam pipelines (out of scope: CPU data pipeline).

Semantics kept from the reference:
  * a strategy is a callable returning the index of the dataset to draw the NEXT batch from;
  * `MultiDataLoader.__next__` pulls from the current loader, restarts an exhausted iterator
    (or, for `should_exhaust_all_iterators` strategies, retires it until all are exhausted), tags
    the batch with `dataset_name` and `task`, then asks the strategy for the next index
    (multi_data_loader.py:121-191);
  * random strategies draw from the GLOBAL NumPy RNG, so identical seeds on all ranks give
    identical task orders (tools/train.py:211-215); `build_iteration_strategy` instantiates the
    strategy twice and burns 300 draws on the second instance (build.py:78-87).
"""
import warnings

import numpy as np

from . import synth


class IterationStrategy:
    name = None

    def __init__(self, dataloaders, **kwargs):
        self.dataloaders = dataloaders

    @property
    def should_exhaust_all_iterators(self):
        return False

    def __call__(self, *args, **kwargs):
        raise NotImplementedError("__call__ hasn't been implemented")


class ConstantIterationStrategy(IterationStrategy):
    name = 'constant'

    def __init__(self, dataloaders, idx=0, **kwargs):
        super().__init__(dataloaders)
        self._idx = idx

    @property
    def should_exhaust_all_iterators(self):
        return True

    def __call__(self, *args, **kwargs):
        return self._idx


class RoundRobinIterationStrategy(IterationStrategy):
    name = 'round_robin'

    def __init__(self, dataloaders, start_idx=0, **kwargs):
        super().__init__(dataloaders)
        self._current_idx = start_idx

    def __call__(self, *args, **kwargs):
        nxt = self._current_idx
        self._current_idx = (self._current_idx + 1) % len(self.dataloaders)
        return nxt


class RepeatedSequenceIterationStrategy(IterationStrategy):
    name = 'repeated_sequence'

    def __init__(self, dataloaders, sequence, **kwargs):
        super().__init__(dataloaders)
        self._current_idx = 0
        assert max(sequence) == len(dataloaders) - 1 and min(sequence) == 0 and \
            len(np.unique(sequence)) == len(dataloaders)
        self.sequence = sequence

    def __call__(self, *args, **kwargs):
        nxt = self._current_idx
        self._current_idx = (self._current_idx + 1) % len(self.sequence)
        return self.sequence[nxt]


class RandomIterationStrategy(IterationStrategy):
    name = 'random'

    def __call__(self, *args, **kwargs):
        return np.random.choice(len(self.dataloaders), 1)[0]


class WeightedRandomIterationStrategy(IterationStrategy):
    name = 'weighted_random'

    def __init__(self, dataloaders, p, **kwargs):
        super().__init__(dataloaders)
        assert len(p) == len(dataloaders)
        p_sum = sum(p)
        # the reference only sets self.p when sum(p) != 1 and then asserts float equality
        # (iteration_strategies.py:192-196); normalising always is the
        # behaviour it intends and is identical whenever the reference does not crash
        self.p = [val / p_sum for val in p]

    def __call__(self, *args, **kwargs):
        return np.random.choice(len(self.dataloaders), 1, p=self.p)[0]


class SizeProportionalIterationStrategy(IterationStrategy):
    name = 'size_proportional'

    def __init__(self, dataloaders, **kwargs):
        super().__init__(dataloaders)
        lengths = []
        for loader in self.dataloaders.values():
            assert hasattr(loader, 'dataset'), "loaders need dataset objects to work with 'size_proportional' sampling"
            n = len(loader.dataset)
            assert n, f'dataset: {loader.dataset} is empty'
            lengths.append(n)
        total = sum(lengths)
        self._dataset_probabilities = [n / total for n in lengths]

    @property
    def should_exhaust_all_iterators(self):
        return True

    def __call__(self, *args, **kwargs):
        return np.random.choice(len(self.dataloaders), 1, p=self._dataset_probabilities)[0]


strategies_map = {
    'constant': ConstantIterationStrategy,
    'round_robin': RoundRobinIterationStrategy,
    'random': RandomIterationStrategy,
    'size_proportional': SizeProportionalIterationStrategy,
    'repeated_sequence': RepeatedSequenceIterationStrategy,
    'weighted_random': WeightedRandomIterationStrategy,
}


def build_iteration_strategy(cfg, data_loaders, verbose=False):
    """mtl/data/build.py:69-88 including the second instance that draws 300 indices."""
    if 'strategy' in cfg:
        kwargs = dict(cfg['strategy'])
        stype = kwargs.pop('type')
    else:
        stype, kwargs = 'round_robin', {}
    cls = strategies_map[stype]
    strategy = cls(data_loaders, **kwargs)
    probe = cls(data_loaders, **kwargs)
    draws = [probe() for _ in range(300)]
    if verbose:
        counts = [draws.count(i) for i in range(len(data_loaders))]
        print(f'Iteration strategy {stype}: first draws {draws[:30]} ratio {counts}')
    return strategy


class MultiDataLoader:
    def __init__(self, loaders, iteration_strategy=None):
        if loaders is None or len(loaders) == 0:
            warnings.warn('Empty loaders passed into MultiDataLoader. This can have unintended consequences.')
        if iteration_strategy is None:
            iteration_strategy = RoundRobinIterationStrategy(loaders)
        self._iteration_strategy = iteration_strategy
        self._loaders = loaders
        self._num_datasets = len(loaders)
        self.dataset_list = list(loaders.keys())
        self._iterators = {}
        self._finished_iterators = {}
        self.current_index = 0
        self.lengths = {name: len(loader) for name, loader in loaders.items()}

    loaders = property(lambda self: self._loaders)
    num_datasets = property(lambda self: self._num_datasets)
    iteration_strategy = property(lambda self: self._iteration_strategy)
    current_dataset_name = property(lambda self: self.dataset_list[self.current_index])
    current_loader = property(lambda self: self._loaders[self.current_dataset_name])
    current_iterator = property(lambda self: self._iterators[self.current_dataset_name])

    @property
    def current_dataset(self):
        return getattr(self.current_loader, 'dataset', None)

    def __len__(self):
        return sum(self.lengths.values())

    def __iter__(self):
        self._finished_iterators = {}
        self._iterators = {key: iter(loader) for key, loader in self._loaders.items()}
        self.change_dataloader()
        return self

    def __next__(self):
        try:
            next_batch = next(self.current_iterator)
        except StopIteration:
            if self.iteration_strategy.should_exhaust_all_iterators:
                self._finished_iterators[self.current_dataset_name] = 1
                if len(self._finished_iterators) == self.num_datasets:
                    raise
                self.change_dataloader()
                next_batch = next(self.current_iterator)
            else:
                self._iterators[self.current_dataset_name] = iter(self.current_loader)
                next_batch = next(self.current_iterator)
        name = self.current_dataset_name
        task = getattr(self.current_dataset, 'task', None)
        self.change_dataloader()
        next_batch['dataset_name'] = name
        next_batch['task'] = task
        return next_batch

    def change_dataloader(self):
        if self.num_datasets <= 1:
            self.current_index = 0
            return
        choice = self.iteration_strategy()
        while self.dataset_list[choice] in self._finished_iterators:
            choice = self.iteration_strategy()
        self.current_index = choice


# ----------------------------------------------------------------------------------------------
# synthetic stand-ins for the per-task datasets / loaders
# ----------------------------------------------------------------------------------------------
class SyntheticDataset:
    def __init__(self, task, length, size, seed=2022):
        self.task, self.length, self.size, self.seed = task, length, size, seed

    def __len__(self):
        return self.length


class SyntheticLoader:
    """Yields pre-generated device-resident batches (a small pool cycled over), so the timed
    region starts with its inputs already in HBM."""

    def __init__(self, dataset, batch_size, device, pool=4, num_batches=None, **batch_kw):
        self.dataset, self.batch_size, self.device = dataset, batch_size, device
        self.num_batches = num_batches or max(len(dataset) // batch_size, 1)
        self.pool = [synth.make_batch(dataset.task, batch_size, dataset.size, seed=dataset.seed + 17 * i,
                                      device=device, **batch_kw) for i in range(pool)]

    def __len__(self):
        return self.num_batches

    def __iter__(self):
        for i in range(self.num_batches):
            b = self.pool[i % len(self.pool)]
            yield dict(b, img_metas=[dict(m) for m in b['img_metas']])


def build_synthetic_multidataloader(cfg, device, size=512, batch_size=2, rank=0, strategy_cfg=None,
                                    lengths=None, pool=4, tasks=None, **batch_kw):
    """One synthetic loader per entry of cfg.data (task taken from the config), wrapped in a
    MultiDataLoader with the configured strategy (default round_robin).  `tasks`: keep only the datasets of these tasks
    (BASELINE configs[3]: the det-only workload); batch_kw goes to synth.make_batch (e.g. max_gt)."""
    loaders = {}
    for i, (name, d) in enumerate(cfg['data'].items()):
        if tasks is not None and d['task'] not in tasks:
            continue
        n = (lengths or {}).get(name, 1 << 30)
        ds = SyntheticDataset(d['task'], n, size, seed=2022 + 1000 * rank + 100 * i)
        loaders[name] = SyntheticLoader(ds, batch_size, device, pool=pool,
                                        num_batches=None if lengths else 1 << 30, **batch_kw)
    scfg = dict(strategy=strategy_cfg) if strategy_cfg else ({'strategy': cfg['strategy']} if 'strategy' in cfg else {})
    return MultiDataLoader(loaders, build_iteration_strategy(scfg, loaders))

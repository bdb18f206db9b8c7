"""Datasets for the propagation + scoring path.

Mirrors the reference's dataset contract (dataset.py:10-14 get_dataset,
:47-64 BasicDataset attributes, :116-131 sampler, :140-164 ProcessedDataset
text format, :258-273 AuxiliaryDataset) with CSR arrays as the primary storage
so the lists can be handed to the device kernels without Python loops.
The raw-dump preprocessors (dataset.py:167-255) are out of scope.
"""
import os
import random
import sys

import numpy as np


def get_dataset(config):
    """Factory by class name, as the reference's get_dataset (dataset.py:10-14)."""
    config = config.copy()
    cls = getattr(sys.modules[__name__], config['name'])
    return cls(config)


def lists_to_csr(lists, sort=False):
    lens = np.fromiter((len(x) for x in lists), dtype=np.int64, count=len(lists))
    rowptr = np.zeros(len(lists) + 1, dtype=np.int64)
    np.cumsum(lens, out=rowptr[1:])
    col = np.fromiter((i for x in lists for i in x), dtype=np.int64, count=int(rowptr[-1]))
    if sort and col.size:
        row = np.repeat(np.arange(len(lists), dtype=np.int64), lens)
        order = np.lexsort((col, row))
        col = col[order]
    return rowptr, col


def csr_to_lists(rowptr, col):
    col = col.tolist()
    return [col[rowptr[u]:rowptr[u + 1]] for u in range(len(rowptr) - 1)]


def _notifying(name):
    def method(self, *args, **kw):
        out = getattr(list, name)(self, *args, **kw)
        self._changed()
        return out
    method.__name__ = name
    return method


class _SplitLists(list):
    """The list-of-lists of one split.  Mutating the OUTER list — dataset.test_data[user] = [...], the reference's
    own idiom (trainer.py:183-216, run/dropui/dataset_dropui.py:13-21) — invalidates the split's cached CSR / device
    views by itself.  Editing an inner list in place (dataset.test_data[user].append(i)) is not seen: call
    dataset.invalidate(split)."""

    def __init__(self, items, changed):
        super().__init__(items)
        self._changed = changed

    for _n in ('__setitem__', '__delitem__', '__iadd__', '__imul__', 'append', 'extend', 'insert', 'pop', 'remove',
               'clear', 'sort', 'reverse'):
        locals()[_n] = _notifying(_n)
    del _n

    def __reduce__(self):                      # pickles / copies as a plain list; BasicDataset._get_list wraps it again
        return (list, (list(self),))


class BasicDataset:
    """Attribute contract of the reference's BasicDataset (dataset.py:47-64).

    train_data / val_data / test_data are properties: assigning a new list-of-lists bumps the split's
    version (what the trainers key their device copies on) and drops its cached CSR views, and so does
    assigning to / deleting from / appending to the list of a split (dataset.test_data[user] = [], as
    trainer.py:183-184 does).  Only code that edits one user's INNER list in place must call invalidate(which).
    One difference from the reference's plain attributes: assignment COPIES the outer list (dataset.test_data = lst;
    lst[u] = [] afterwards is not seen — edit dataset.test_data[u] instead).  Tracking survives pickle / deepcopy."""

    SPLITS = ('train', 'val', 'test')

    def __init__(self, dataset_config):
        self.config = dataset_config
        self.name = dataset_config['name']
        self.device = dataset_config.get('device', 'cpu')
        self.negative_sample_ratio = dataset_config.get('neg_ratio', 1)
        self.n_users = 0
        self.n_items = 0
        self.train_array = None
        self._lists = {}
        self._csr = {}
        self._version = dict.fromkeys(self.SPLITS, 0)

    # ---- list views <-> CSR views ------------------------------------------------
    def _get_list(self, name):
        cur = self._lists.get(name)
        if cur is not None and not isinstance(cur, _SplitLists):
            # a dataset that went through pickle / deepcopy / mp.spawn carries its splits as plain lists (_SplitLists
            # reduces to one: its callback is bound to the ORIGINAL dataset): track them again on first access
            cur = self._lists[name] = self._track(name, cur)
        return cur

    def _track(self, name, value):
        return _SplitLists(value, lambda: self.invalidate(name))

    def _set_list(self, name, value):
        self._lists[name] = self._track(name, value)
        self._drop_csr(name)
        self._version[name] += 1

    def _drop_csr(self, which):
        for k in [k for k in self._csr if which is None or k[0] == which]:
            del self._csr[k]

    train_data = property(lambda self: self._get_list('train'), lambda self, v: self._set_list('train', v))
    val_data = property(lambda self: self._get_list('val'), lambda self, v: self._set_list('val', v))
    test_data = property(lambda self: self._get_list('test'), lambda self, v: self._set_list('test', v))

    def version(self, which):
        """Counter of the changes made to a split's lists (assignment or invalidate())."""
        return self._version[which]

    def csr(self, which, sort=True):
        """(rowptr int64, col int64) of train/val/test lists; `sort` orders each
        user's items ascending (what the device membership tests expect)."""
        key = (which, sort)
        if key not in self._csr:
            self._csr[key] = lists_to_csr(getattr(self, which + '_data'), sort=sort)
        return self._csr[key]

    def invalidate(self, which=None):
        """Call after mutating train/val/test lists in place; `which` limits it to one split."""
        self._drop_csr(which)
        for name in self.SPLITS:
            if which is None or name == which:
                self._version[name] += 1

    def _finish(self):
        rowptr, col = lists_to_csr(self.train_data)
        users = np.repeat(np.arange(self.n_users, dtype=np.int64), np.diff(rowptr))
        self.train_array = np.stack([users, col], axis=1)

    def __len__(self):
        return len(self.train_array)

    def __getitem__(self, index):
        """negative_sample_ratio BPR draws for one random user, int64 [ratio, 3]; `index` is ignored — the
        sampling protocol of dataset.py:119-131 (uniform user among those with train items, one uniform
        positive shared by the draws, negatives rejected while they are train items of the user).  Kept for
        DataLoader-style callers; the trainers draw on the device (igcn_bpr_sample)."""
        lists = self.train_data
        while True:
            user = random.randint(0, self.n_users - 1)
            if lists[user]:
                break
        seen = lists[user]
        draws = np.empty((self.negative_sample_ratio, 3), dtype=np.int64)
        draws[:, 0] = user
        draws[:, 1] = np.random.choice(seen)
        for row in draws:
            candidate = random.randint(0, self.n_items - 1)
            while candidate in seen:
                candidate = random.randint(0, self.n_items - 1)
            row[2] = candidate
        return draws

    def sample_batch_host(self, batch_size, rng):
        """Vectorised host sampler with the distribution of __getitem__:
        uniform non-empty user, uniform positive, rejection-sampled negative.
        Returns int64 [batch, 3]."""
        rowptr, col = self.csr('train', sort=True)
        lens = np.diff(rowptr)
        nonempty = np.flatnonzero(lens > 0)
        users = nonempty[rng.integers(0, nonempty.size, size=batch_size)]
        pos = col[rowptr[users] + (rng.random(batch_size) * lens[users]).astype(np.int64)]
        neg = rng.integers(0, self.n_items, size=batch_size)
        key = users * np.int64(self.n_items)
        allkeys = np.repeat(np.arange(self.n_users, dtype=np.int64), lens) * np.int64(self.n_items) + col
        todo = np.arange(batch_size)
        while todo.size:
            k = key[todo] + neg[todo]
            pos_in = np.searchsorted(allkeys, k)
            hit = (pos_in < allkeys.size) & (allkeys[np.minimum(pos_in, allkeys.size - 1)] == k)
            todo = todo[hit]
            neg[todo] = rng.integers(0, self.n_items, size=todo.size)
        return np.stack([users, pos, neg], axis=1).astype(np.int64)

    def output_dataset(self, path):
        """Writes train/val/test.txt in the reference format (dataset.py:40-44, :133-137)."""
        os.makedirs(path, exist_ok=True)
        for name in ('train', 'val', 'test'):
            with open(os.path.join(path, name + '.txt'), 'w') as f:
                for user, items in enumerate(getattr(self, name + '_data')):
                    f.write(' '.join([str(user)] + [str(i) for i in items]) + '\n')


class ProcessedDataset(BasicDataset):
    """Reader of the reference's processed text format (dataset.py:140-164):
    one line per user, 'user item item ...'; n_items = max id + 1 over all files."""

    def __init__(self, dataset_config):
        super().__init__(dataset_config)
        path = dataset_config['path']
        cache = os.path.join(path, BINARY_SPLITS)
        texts = [os.path.join(path, f) for f in ('train.txt', 'val.txt', 'test.txt')]
        if dataset_config.get('binary_cache') and os.path.exists(cache) and \
                all(os.path.getmtime(cache) >= os.path.getmtime(t) for t in texts if os.path.exists(t)):
            self.n_users, self.n_items, csrs = _load_binary(path)       # parsed once before: skip the text
            for name, (rp, col) in csrs.items():
                setattr(self, name + '_data', csr_to_lists(rp, col))
                self._csr[(name, False)] = (rp, col)
            self._finish()
            return
        self.train_data = self.read_data(os.path.join(dataset_config['path'], 'train.txt'))
        self.val_data = self.read_data(os.path.join(dataset_config['path'], 'val.txt'))
        self.test_data = self.read_data(os.path.join(dataset_config['path'], 'test.txt'))
        assert len(self.train_data) == len(self.val_data)
        assert len(self.train_data) == len(self.test_data)
        self.n_users = len(self.train_data)
        self._finish()
        if dataset_config.get('binary_cache'):
            try:
                save_binary(self, path)
            except OSError:
                pass                                                     # read-only data directory: no cache

    def read_data(self, file_path):
        """One list of item ids per line ('user item item ...', dataset.py:154-164); the user id in the
        first column is positional and not read; n_items grows to the largest id seen + 1."""
        with open(file_path, 'r') as f:
            rows = [line.split(' ')[1:] for line in f.read().strip().split('\n')]
        data = [list(map(int, fields)) for fields in rows]
        largest = max((max(items) for items in data if items), default=-1)
        self.n_items = max(self.n_items, largest + 1)
        return data


BINARY_SPLITS = 'splits_csr_v1.npz'


def save_binary(dataset, path):
    """Writes the train/val/test lists as CSR arrays (one .npz): the on-disk form for splits too large
    for the text format (BASELINE config 5: 500 M pairs), read back by BinaryDataset and used as a
    cache by ProcessedDataset(binary_cache=True).  List order is kept (training samples positives by
    position, dataset.py:124)."""
    os.makedirs(path, exist_ok=True)
    arrays = {'n_users': np.int64(dataset.n_users), 'n_items': np.int64(dataset.n_items)}
    for name in ('train', 'val', 'test'):
        rp, col = dataset.csr(name, sort=False)
        arrays[name + '_rowptr'], arrays[name + '_col'] = rp, col
    tmp = os.path.join(path, BINARY_SPLITS + '.tmp.npz')
    np.savez(tmp, **arrays)
    os.replace(tmp, os.path.join(path, BINARY_SPLITS))


def _load_binary(path):
    with np.load(os.path.join(path, BINARY_SPLITS)) as z:
        csrs = {name: (z[name + '_rowptr'], z[name + '_col']) for name in ('train', 'val', 'test')}
        return int(z['n_users']), int(z['n_items']), csrs


class CsrBackedDataset(BasicDataset):
    """A dataset whose train/val/test lists are stored as CSR arrays; the Python
    list-of-lists views of the reference contract (dataset.train_data[user], ...)
    are materialised lazily and may be re-assigned (inductive_eval does)."""

    def __init__(self, dataset_config, n_users=0, n_items=0, csrs=None):
        super().__init__(dataset_config)
        if csrs is not None:
            self._install(n_users, n_items, csrs)

    def _install(self, n_users, n_items, csrs):
        """csrs: {'train'|'val'|'test': (rowptr int64 [n_users+1], col int64)} in list order."""
        self.n_users, self.n_items = int(n_users), int(n_items)
        self._lists = {}
        self._csr = {(name, False): (np.asarray(rp, dtype=np.int64), np.asarray(col, dtype=np.int64))
                     for name, (rp, col) in csrs.items()}
        for name in self.SPLITS:
            self._version[name] += 1
        rp, col = self._csr[('train', False)]
        self.train_array = np.stack([np.repeat(np.arange(self.n_users, dtype=np.int64), np.diff(rp)), col], axis=1)

    def _get_list(self, name):
        if name not in self._lists:
            self._lists[name] = self._track(name, csr_to_lists(*self._csr[(name, False)]))
        return self._lists[name]

    def csr(self, which, sort=True):
        key = (which, sort)
        if key not in self._csr:
            if (which, False) in self._csr and which not in self._lists:
                rp, col = self._csr[(which, False)]
                row = np.repeat(np.arange(self.n_users, dtype=np.int64), np.diff(rp))
                self._csr[key] = (rp, col[np.lexsort((col, row))])
            else:
                self._csr[key] = lists_to_csr(self._get_list(which), sort=sort)
        return self._csr[key]

    def invalidate(self, which=None):
        """The CSR arrays are the primary storage: only the splits whose LISTS were materialised (and may
        have been edited) are rebuilt from them."""
        for name in self.SPLITS:
            if (which is None or name == which) and name in self._lists:
                self._drop_csr(name)
                self._version[name] += 1


class BinaryDataset(CsrBackedDataset):
    """Splits stored by save_binary() (config: {'name': 'BinaryDataset', 'path': dir}); the list views of
    the reference contract are materialised only if something asks for them."""

    def __init__(self, dataset_config):
        super().__init__(dataset_config)
        n_users, n_items, csrs = _load_binary(dataset_config['path'])
        self._install(n_users, n_items, csrs)


def _filter_csr(rowptr, col, keep, n_rows):
    """Rows [0, n_rows) of a CSR, entries where `keep` is True (order preserved)."""
    rows = np.repeat(np.arange(len(rowptr) - 1, dtype=np.int64), np.diff(rowptr))
    m = keep & (rows < n_rows)
    rp = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows[m], minlength=n_rows)[:n_rows], out=rp[1:])
    return rp, col[m]


def dropit_dataset(dataset, ratio, name=None):
    """Each user's train list cut to its first int(len * ratio) items; val / test unchanged
    (the reference's run/dropit/dataset_dropit.py:6-9).  Returns a new dataset."""
    csrs = {}
    for which in ('train', 'val', 'test'):
        rp, col = dataset.csr(which, sort=False)
        if which == 'train':
            lens = np.diff(rp)
            pos = np.arange(col.shape[0], dtype=np.int64) - np.repeat(rp[:-1], lens)
            keep = pos < np.repeat((lens * ratio).astype(np.int64), lens)
            rp, col = _filter_csr(rp, col, keep, dataset.n_users)
        csrs[which] = (rp, col)
    return CsrBackedDataset({'name': name or dataset.name + '_dropit', 'device': dataset.device},
                            dataset.n_users, dataset.n_items, csrs)


def resize_dataset(dataset, ratio, name=None):
    """The first int(n_users * ratio) users and int(n_items * ratio) items, all three lists
    filtered (the reference's run/dropui/dataset_dropui.py:7-29).  Returns a new dataset."""
    n_users, n_items = int(dataset.n_users * ratio), int(dataset.n_items * ratio)
    csrs = {}
    for which in ('train', 'val', 'test'):
        rp, col = dataset.csr(which, sort=False)
        csrs[which] = _filter_csr(rp, col, col < n_items, n_users)
    return CsrBackedDataset({'name': name or dataset.name + '_dropui', 'device': dataset.device}, n_users, n_items, csrs)


class SyntheticDataset(CsrBackedDataset):
    """Seeded synthetic implicit-feedback split with the shape of the paper's
    datasets: per-user interaction counts ~ max(min_inter,
    LogNormal), items drawn from a Zipf-Mandelbrot popularity over a random
    permutation (zipf_q = 0 gives the pure Zipf stress case, zipf_a = 0 the
    uniform one), de-duplicated per user, split 70/10/20 per user in draw order
    as dataset.py:94-114 does.

    communities = C > 1 (round 4, measurement only — the headline stays on the graph above): PLANTED user / item communities of
    equal size and the same degree laws.  The item at popularity rank r belongs to community r mod C (every community gets
    the same popularity profile), a user to a uniformly drawn one; a draw of rank r goes, with probability community_share,
    to the item of (nearly) the same rank in the user's OWN community instead.  Item ids stay a random permutation of the
    ranks, user ids carry no information: the structure is in the graph, not in the labels.  The planted labels are kept
    as user_community / item_community (int64) for the experiments that need the ground truth."""

    PRESETS = {
        'gowalla': dict(n_users=29858, n_items=40988, n_inter=1027464),
        'yelp': dict(n_users=75173, n_items=42706, n_inter=1931173),
        'amazon': dict(n_users=109730, n_items=96421, n_inter=3181759),
    }

    def __init__(self, dataset_config):
        super().__init__(dataset_config)
        cfg = dict(self.PRESETS.get(dataset_config.get('preset', ''), {}))
        cfg.update({k: dataset_config[k] for k in ('n_users', 'n_items', 'n_inter') if k in dataset_config})
        self.n_users, self.n_items = int(cfg['n_users']), int(cfg['n_items'])
        n_inter = int(cfg['n_inter'])
        seed = dataset_config.get('seed', 2021)
        min_inter = dataset_config.get('min_inter', 10)
        zipf_a = dataset_config.get('zipf_a', 1.0)
        zipf_q = dataset_config.get('zipf_q', 150.0)
        split = dataset_config.get('split_ratio', [0.7, 0.1, 0.2])
        rng = np.random.default_rng(seed)

        # per-user counts: max(min_inter, lognormal), scale found by bisection on the total
        z = rng.standard_normal(self.n_users)
        lo, hi = -5., 12.
        for _ in range(60):
            mu = 0.5 * (lo + hi)
            cnt = np.maximum(min_inter, np.rint(np.exp(mu + z))).astype(np.int64)
            if cnt.sum() > n_inter:
                hi = mu
            else:
                lo = mu
        cnt = np.minimum(cnt, self.n_items // 2)
        # popularity over a random permutation of the items
        rank = np.arange(1, self.n_items + 1, dtype=np.float64)
        p = 1. / np.power(rank + zipf_q, zipf_a)
        cdf = np.cumsum(p / p.sum())
        perm = rng.permutation(self.n_items)

        users = np.repeat(np.arange(self.n_users, dtype=np.int64), cnt)
        ranks = np.minimum(np.searchsorted(cdf, rng.random(users.size)), self.n_items - 1)
        n_comm = int(dataset_config.get('communities', 0) or 0)
        if n_comm > 1:
            crng = np.random.default_rng([seed, 77])                   # a stream of its own: communities = 0 draws as before
            self.user_community = crng.integers(0, n_comm, self.n_users)
            own = crng.random(users.size) < float(dataset_config.get('community_share', 0.8))
            in_own = (ranks // n_comm) * n_comm + self.user_community[users]
            ranks = np.where(own & (in_own < self.n_items), in_own, ranks)
            self.item_community = np.empty(self.n_items, dtype=np.int64)
            self.item_community[perm] = np.arange(self.n_items) % n_comm
        items = perm[ranks]
        # de-duplicate per user, keeping draw order (first occurrence)
        key = users * np.int64(self.n_items) + items
        _, first = np.unique(key, return_index=True)
        first.sort()
        users, items = users[first], items[first]
        cnt = np.bincount(users, minlength=self.n_users)
        rowptr = np.zeros(self.n_users + 1, dtype=np.int64)
        np.cumsum(cnt, out=rowptr[1:])
        n_train = (cnt * split[0]).astype(np.int64)
        n_test = (cnt * split[2]).astype(np.int64)
        pos_in_user = np.arange(users.size, dtype=np.int64) - rowptr[users]
        is_train = pos_in_user < n_train[users]
        is_test = pos_in_user >= (cnt - n_test)[users]
        is_test &= n_test[users] > 0
        is_val = ~is_train & ~is_test
        csrs = {}
        for name, m in (('train', is_train), ('val', is_val), ('test', is_test)):
            c = np.bincount(users[m], minlength=self.n_users)
            rp = np.zeros(self.n_users + 1, dtype=np.int64)
            np.cumsum(c, out=rp[1:])
            csrs[name] = (rp, items[m].astype(np.int64))
        self._install(self.n_users, self.n_items, csrs)


class AuxiliaryDataset(BasicDataset):
    """Train lists re-indexed into template-id space for the INMO auxiliary
    loss (dataset.py:258-273).  user_map / item_map: dict or None (= identity)."""

    def __init__(self, dataset, user_map, item_map):
        super().__init__({'name': 'AuxiliaryDataset', 'device': dataset.device})
        self.n_users = dataset.n_users if user_map is None else len(user_map)
        self.n_items = dataset.n_items if item_map is None else len(item_map)
        self.negative_sample_ratio = 1
        self.length = len(dataset)
        rowptr, col = dataset.csr('train', sort=False)
        users = np.repeat(np.arange(dataset.n_users, dtype=np.int64), np.diff(rowptr))
        u_lut = np.arange(dataset.n_users, dtype=np.int64)
        i_lut = np.arange(dataset.n_items, dtype=np.int64)
        if user_map is not None:
            u_lut = np.full(dataset.n_users, -1, dtype=np.int64)
            for k, v in user_map.items():
                if k < dataset.n_users:
                    u_lut[k] = v
        if item_map is not None:
            i_lut = np.full(dataset.n_items, -1, dtype=np.int64)
            for k, v in item_map.items():
                if k < dataset.n_items:
                    i_lut[k] = v
        tu, ti = u_lut[users], i_lut[col]
        keep = (tu >= 0) & (ti >= 0)
        tu, ti = tu[keep], ti[keep]
        order = np.argsort(tu, kind='stable')          # original users ascending within a template user
        tu, ti = tu[order], ti[order]
        rp = np.zeros(self.n_users + 1, dtype=np.int64)
        np.cumsum(np.bincount(tu, minlength=self.n_users), out=rp[1:])
        self.train_data = csr_to_lists(rp, ti)
        self.train_array = np.stack([tu, ti], axis=1)

    def __len__(self):
        return self.length

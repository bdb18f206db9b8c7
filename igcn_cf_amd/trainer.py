"""Trainers: epoch loop, BPR / INMO losses, evaluation and metrics.

Host-side mirror of the reference's trainer contract (trainer.py:14-20
get_trainer, :23-219 BasicTrainer, :222-248 BPRTrainer, :281-320 IGCNTrainer):
same class names, config keys, method names and return values.  The hot loops
run on the device:

* train step: device-side negative sampling (igcn_bpr_sample) -> K-layer SpMM
  propagation -> fused BPR gather/dot/softplus/L2 -> backward through the same
  kernels -> torch Adam (trainer.py:231-248, :294-320);
* eval: propagation ONCE, then fused score + mask + top-k over all users
  (trainer.py:140-164 does predict/mask/topk per 512-user batch with a Python
  exclusion loop), hit matrix on the device (trainer.py:111-115), the final
  Precision / Recall / NDCG reductions in numpy exactly as trainer.py:116-137.

Baseline trainers outside the hot path (IDCFTrainer, BCETrainer, MLTrainer) are
out of scope.
"""
import functools
import os
import sys
import time

import numpy as np
import torch
from torch.optim import SGD, Adam  # noqa: F401  (resolved by name, trainer.py:43-45)

from . import ops
from ._lib import MAX_METRIC_CUTS, MAX_TOPK
from .dataset import AuxiliaryDataset


def get_trainer(config, dataset, model):
    """Factory by class name (trainer.py:14-20)."""
    config = config.copy()
    config['dataset'] = dataset
    config['model'] = model
    cls = getattr(sys.modules[__name__], config['name'])
    return cls(config)


class AverageMeter:
    def __init__(self):
        self.avg = 0.
        self.sum = 0.
        self.count = 0.

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


_M64 = 0xFFFFFFFFFFFFFFFF


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def batch_seed(seed, call):
    """64-bit seed of the call-th batch: both words change unpredictably from one batch to the next, so the
    counter hash of the device sampler (csrc/common.h hash_counter XORs the low seed word into the draw
    counter) cannot line up draw i of one batch with draw i^1 of the next."""
    return _splitmix64((seed & _M64) ^ _splitmix64(call & _M64))


def _csr_to_device(rowptr, col, device):
    return (torch.from_numpy(np.ascontiguousarray(rowptr, dtype=np.int64)).to(device),
            torch.from_numpy(np.ascontiguousarray(col, dtype=np.int32)).to(device))


def _sorted_csr_device(csrs, n_items, device):
    """Per-user union (as a multiset) of one or more host CSR lists (any order inside a user), each user's items
    sorted ascending, as device arrays (rowptr int64, col int32): the sort runs on the GPU (one torch.sort of
    user * n_items + item keys), not as a host lexsort per evaluation stage."""
    n = len(csrs[0][0]) - 1
    counts = sum(torch.from_numpy(np.diff(rp).astype(np.int64)) for rp, _ in csrs)
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    torch.cumsum(counts, 0, out=rowptr[1:])
    keys = []
    users = torch.arange(n, dtype=torch.int64, device=device)
    for rp, col in csrs:
        lens = torch.from_numpy(np.diff(rp).astype(np.int64)).to(device)
        c = torch.from_numpy(np.ascontiguousarray(col, dtype=np.int64)).to(device)
        keys.append(torch.repeat_interleave(users, lens) * n_items + c)
    key = torch.sort(torch.cat(keys) if len(keys) > 1 else keys[0]).values
    return rowptr.to(device), (key % n_items).to(torch.int32)


def _merge_sorted_csr(a, b):
    """Per-user union (as a multiset) of two CSR lists, each row sorted ascending."""
    (rp_a, col_a), (rp_b, col_b) = a, b
    n = len(rp_a) - 1
    la, lb = np.diff(rp_a), np.diff(rp_b)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(la + lb, out=rowptr[1:])
    rows = np.concatenate([np.repeat(np.arange(n, dtype=np.int64), la), np.repeat(np.arange(n, dtype=np.int64), lb)])
    cols = np.concatenate([col_a, col_b])
    order = np.lexsort((cols, rows))
    return rowptr, cols[order]


class DeviceSampler:
    """BPR triplet batches drawn on the GPU from the train CSR (replaces the
    DataLoader over BasicDataset.__getitem__, trainer.py:226-227)."""

    def __init__(self, dataset, device, seed):
        rowptr, col = dataset.csr('train', sort=False)
        self.rowptr, self.col = _sorted_csr_device([(rowptr, col)], dataset.n_items, device)
        nonempty = np.flatnonzero(np.diff(rowptr) > 0).astype(np.int32)
        self.nonempty = torch.from_numpy(nonempty).to(device)
        self.n_items = dataset.n_items
        self.length = len(dataset)
        self.seed = int(seed)
        self.calls = 0

    def epoch_batches(self, batch_size, into=None):
        """ceil(len / batch_size) batches, the last one short, as a DataLoader over
        a dataset of len(train_array) draws (dataset.py:116-117).  into: callable(b) -> tensor to draw batch b into, or
        None (a captured step's input buffer: the draw then needs no copy; the yielded tensor is only valid until the
        next draw)."""
        left = self.length
        while left > 0:
            b = min(batch_size, left)
            left -= b
            self.calls += 1
            yield ops.bpr_sample(self.rowptr, self.col, self.nonempty, self.n_items, b, batch_seed(self.seed, self.calls),
                                 out=into(b) if into else None)

    def epoch_node_batches(self, batch_size, item_offset, into=None):
        """The same draws as epoch_batches (same seeds), each as int64 [3 * b] NODE ids: users | item_offset +
        positives | item_offset + negatives — what a graph model's bpr_loss_terms_nodes takes, with no transpose,
        offset add or concatenation per step."""
        left = self.length
        while left > 0:
            b = min(batch_size, left)
            left -= b
            self.calls += 1
            yield ops.bpr_sample_nodes(self.rowptr, self.col, self.nonempty, self.n_items, b,
                                       batch_seed(self.seed, self.calls), item_offset, out=into(b) if into else None)


class BasicTrainer:
    def __init__(self, trainer_config):
        self.config = trainer_config
        self.name = trainer_config['name']
        self.dataset = trainer_config['dataset']
        self.model = trainer_config['model']
        self.topks = trainer_config['topks']
        # limits of the fused scorer (csrc/score_topk.hip), checked here rather than at the first eval
        d = getattr(self.model, 'embedding_size', None)
        if max(self.topks) > MAX_TOPK or (d is not None and (d % 4 or d > 256)):
            raise ValueError('fused score/top-k kernel: needs max(topks) <= %d and embedding_size %% 4 == 0, <= 256 '
                             '(got topks=%s, embedding_size=%s)' % (MAX_TOPK, self.topks, d))
        self.device = torch.device(trainer_config['device'])
        self.n_epochs = trainer_config['n_epochs']
        self.max_patience = trainer_config.get('max_patience', 50)
        self.val_interval = trainer_config.get('val_interval', 1)
        self.test_batch_size = trainer_config.get('test_batch_size', 512)
        # users scored per fused launch; the reference's test_batch_size (512) only
        # bounds its dense score block and does not change results
        self.eval_chunk = trainer_config.get('eval_chunk', 1 << 18)
        self.epoch = 0
        self.best_ndcg = -np.inf
        self.save_path = None
        self.opt = None
        self._excl_cache = {}

    def initialize_optimizer(self):
        opt = getattr(sys.modules[__name__], self.config['optimizer'])
        kw = {}
        if opt is Adam and self.config.get('fused_optimizer', True):
            kw['fused'] = True             # same update rule, one kernel over all parameters
            if self.config.get('hip_graph', True):
                kw['capturable'] = True    # the step count lives on the device, so the step can sit in a HIP graph
        self.opt = opt(self.model.parameters(), lr=self.config['lr'], **kw)

    # ---- one optimisation step as ONE captured HIP graph (config key 'hip_graph', default True) -----------------
    def _graph_wanted(self):
        """Steps are captured unless the config says 'hip_graph': False or the optimizer cannot sit in a graph: torch's
        Adam reads its step count on the host unless built capturable (initialize_optimizer does that); SGD has no host
        state.  Decided up front — nothing raised during a capture is ever swallowed."""
        if not self.config.get('hip_graph', True) or self.opt is None:
            return False
        if isinstance(self.opt, Adam):
            return all(g.get('capturable', False) for g in self.opt.param_groups)
        return isinstance(self.opt, SGD)

    _tokens = iter(range(1, 1 << 62))

    @classmethod
    def _token(cls, obj):
        """A number that identifies `obj` for good (id() can come back after the object is freed)."""
        if obj is None:
            return 0
        tok = getattr(obj, '_igcn_token', None)
        if tok is None:
            tok = next(cls._tokens)
            obj._igcn_token = tok
        return tok

    def _graph_key(self):
        """Everything a captured step bakes in as pointers or launch constants: a swapped graph / feature matrix /
        optimizer, or a changed dropout rate, regulariser weight or learning rate, captures again."""
        m = self.model
        return (self._token(getattr(m, 'norm_adj', None)), self._token(getattr(m, 'feat_mat', None)), self._token(self.opt),
                self.batch_size, getattr(m, 'dropout', None), bool(m.training), getattr(self, 'l2_reg', None),
                getattr(self, 'aux_reg', None), tuple(float(g['lr']) for g in self.opt.param_groups))

    def _graph_step(self, inputs, loss_fn, kind='nodes'):
        """Replays the captured step on `inputs` (tensors copied into the static buffers the graph reads); captures
        it first when there is none for the current model / trainer state.  Returns the loss tensor.  An error raised
        while capturing is the caller's: the parameters, the optimizer state and the seeds are put back first."""
        key = self._graph_key() + (kind,)
        if getattr(self, '_graph', None) is None or self._graph_for != key:
            self._graph = None
            self._capture_step(inputs, loss_fn)
            self._graph_for = key
        for st, t in zip(self._static_inputs, inputs):
            if st.data_ptr() != t.data_ptr():                # (a sampler that drew straight into the buffer: nothing to copy)
                st.copy_(t)
        self._graph.replay()
        return self._static_loss.clone()

    def _draw_into(self, index, shape):
        """For the samplers: the captured step's input buffer `index` if a full-size batch of `shape` fits it, else None."""
        def pick(b):
            st = getattr(self, '_static_inputs', None)
            if getattr(self, '_graph', None) is None or st is None or index >= len(st) or not self._graph_wanted():
                return None
            want = tuple(shape(b))
            return st[index] if tuple(st[index].shape) == want else None
        return pick

    def _seed_for_step(self, graph):
        """Dropout seeds of a model that drops edges (IGCN / IMF): a captured step must read its seed from device
        memory — a host seed would be baked into the graph and every replay would drop the same edges — so the seed
        moves there and is advanced before EVERY step, captured or not, from the same CPU-generator sequence the
        eager path draws (model.py:263-267 draws per get_rep call)."""
        m = self.model
        if hasattr(m, 'advance_dropout_seed'):
            if graph:
                m.use_device_seed()
            m.advance_dropout_seed()

    def _capture_step(self, inputs, loss_fn):
        """Warm-up steps (lazy initialisation, workspaces) on a side stream, then the capture.  The warm-up is made
        invisible: parameters, optimizer state, the CPU generator the dropout seeds come from and the device seed are
        put back — also when the warm-up or the capture raises — so a run with hip_graph follows the run without it."""
        self._static_inputs = [t.clone() for t in inputs]
        params = [p for g in self.opt.param_groups for p in g['params']]
        saved_p = [p.detach().clone() for p in params]
        saved_s = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.opt.state[p].items()}
                   for p in params if p in self.opt.state}
        rng = torch.get_rng_state()
        seed_dev = getattr(self.model, '_seed_dev', None)
        saved_seed = seed_dev.clone() if seed_dev is not None else None

        def put_back():
            with torch.no_grad():
                for p, sp in zip(params, saved_p):
                    p.copy_(sp)
                for p in params:
                    for k, v in self.opt.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            old = saved_s.get(id(p), {}).get(k)
                            v.copy_(old) if old is not None else v.zero_()
                if saved_seed is not None:
                    seed_dev.copy_(saved_seed)
            torch.set_rng_state(rng)
            self.opt.zero_grad(set_to_none=True)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(3):
                    if hasattr(self.model, 'advance_dropout_seed'):
                        self.model.advance_dropout_seed()
                    loss = loss_fn(*self._static_inputs)
                    self.opt.zero_grad()
                    loss.backward()
                    self.opt.step()
                    del loss                  # no autograd graph of the warm-up (its AccumulateGrad nodes) outlives it
        finally:
            torch.cuda.current_stream().wait_stream(side)
            put_back()
        graph = torch.cuda.CUDAGraph()
        one = self._graph_one = torch.ones((), dtype=torch.float32, device=self.device)   # the root gradient: autograd would fill a new one per replay
        try:
            with torch.cuda.graph(graph):
                loss = loss_fn(*self._static_inputs)
                loss.backward(one)
                self.opt.step()
                self._static_loss = loss.detach()
        except BaseException:
            torch.cuda.synchronize()
            put_back()
            raise
        self._graph = graph

    def train_one_epoch(self):
        raise NotImplementedError

    def record(self, writer, stage, metrics):
        """tensorboard-style scalars '<model>_<trainer>/<stage>_<metric>@k' (trainer.py:50-55)."""
        prefix = f'{self.model.name}_{self.name}/{stage}_'
        for metric, by_k in metrics.items():
            for k in self.topks:
                writer.add_scalar(f'{prefix}{metric}@{k:d}', by_k[k], self.epoch)

    def _checkpoint_if_best(self, ndcg):
        """Keep only the best-validation-NDCG checkpoint on disk (trainer.py:91-100); returns True
        when `ndcg` is a new best."""
        if ndcg <= self.best_ndcg:
            return False
        if self.save_path:
            os.remove(self.save_path)
        fname = f'{self.model.name}_{self.name}_{self.dataset.name}_{ndcg * 100:.3f}.pth'
        self.save_path = os.path.join('checkpoints', fname)
        self.best_ndcg = ndcg
        self.model.save(self.save_path)
        print(f'Best NDCG, save model to {self.save_path}')
        return True

    def train(self, verbose=True, writer=None):
        """Training driver with the reference's protocol (trainer.py:57-107): per epoch one pass of
        train_one_epoch and an evaluation on the train lists; every val_interval epochs a validation,
        whose NDCG@topks[0] decides checkpointing and early stopping (max_patience epochs without
        improvement); the best checkpoint is reloaded at the end.  A non-trainable model is only
        validated."""
        k0 = self.topks[0]
        if not self.model.trainable:
            summary, metrics = self.eval('val')
            if verbose:
                print(f'Validation result. {summary}')
            return metrics['NDCG'][k0]

        os.makedirs('checkpoints', exist_ok=True)
        tag = f'{self.model.name}_{self.name}'
        epochs_left = self.max_patience
        for self.epoch in range(self.n_epochs):
            t0 = time.time()
            self.model.train()
            loss = self.train_one_epoch()
            _, train_metrics = self.eval('train')
            if verbose:
                print(f'Epoch {self.epoch:d}/{self.n_epochs:d}, Loss: {loss:.6f}, Time: {time.time() - t0:.3f}s')
            if writer:
                writer.add_scalar(f'{tag}/train_loss', loss, self.epoch)
                self.record(writer, 'train', train_metrics)
            if (self.epoch + 1) % self.val_interval:
                continue

            t0 = time.time()
            summary, val_metrics = self.eval('val')
            if verbose:
                print(f'Validation result. {summary}Time: {time.time() - t0:.3f}s')
            if writer:
                self.record(writer, 'validation', val_metrics)
            if self._checkpoint_if_best(val_metrics['NDCG'][k0]):
                epochs_left = self.max_patience
                continue
            epochs_left -= self.val_interval
            if epochs_left <= 0:
                print('Early stopping!')
                break
        self.model.load(self.save_path)
        return self.best_ndcg

    # ---- metrics ---------------------------------------------------------------
    def _metrics_from_hits(self, hit_matrix, eval_data_len):
        """Precision / Recall / NDCG @k from a 0/1 hit matrix, with the array dtypes of trainer.py:116-137
        (float32 hits and discounts, int32 list lengths, hence a float64 recall) so that the values are
        bit-identical to the reference's; users without evaluation items are left out of the means."""
        out = {'Precision': {}, 'Recall': {}, 'NDCG': {}}
        lens = np.asarray(eval_data_len)
        for k in self.topks:
            top = hit_matrix[:, :k]
            n_hit = top.sum(axis=1)
            best_possible = np.minimum(lens, k)
            scored = best_possible > 0
            discount = np.log2(np.arange(2, k + 2, dtype=np.float32))[None, :]
            ideal = (np.arange(k)[None, :] < best_possible[:, None]).astype(np.float32)
            with np.errstate(invalid='ignore', divide='ignore'):
                recall = n_hit / lens
                ndcg = (top / discount).sum(axis=1) / (ideal / discount).sum(axis=1)
            out['Precision'][k] = (n_hit / k)[scored].mean()
            out['Recall'][k] = recall[scored].mean()
            out['NDCG'][k] = ndcg[scored].mean()
        return out

    def _metrics_device(self, rec, eval_rowptr, eval_col):
        """trainer.py:109-138 as one fused pass over the recommended lists (igcn_eval_metrics): per user in float32 as the
        reference forms them, the mean over users in float64; only 3 * len(topks) + 1 numbers cross PCIe."""
        sums, n_valid = ops.eval_metric_sums(rec, eval_rowptr, eval_col, self.topks)
        results = {'Precision': {}, 'Recall': {}, 'NDCG': {}}
        with np.errstate(invalid='ignore', divide='ignore'):
            for i, k in enumerate(self.topks):
                for j, name in enumerate(('Precision', 'Recall', 'NDCG')):
                    results[name][k] = np.float32(np.float64(sums[i, j]) / np.float64(n_valid))   # no evaluated user: nan
        return results

    def _metrics_from_hits_device(self, hit, eval_len):
        """The reductions of trainer.py:116-137 on the device (float32, same formulas): only the
        final scalars cross PCIe.  Used by eval(); calculate_metrics() keeps the numpy path, whose
        results are bit-identical to the reference's."""
        results = {'Precision': {}, 'Recall': {}, 'NDCG': {}}
        lens = eval_len.to(torch.float32)
        valid = eval_len > 0
        n_valid = valid.sum().to(torch.float32)                  # no evaluated user: 0/0 = nan, as the reference's empty mean
        out = []
        for k in self.topks:
            h = hit[:, :k]
            hit_num = h.sum(dim=1)
            denom = torch.log2(torch.arange(2, k + 2, dtype=torch.float32, device=hit.device))
            dcg = (h / denom).sum(dim=1)
            ideal = (torch.arange(k, device=hit.device)[None, :] < eval_len.clamp(max=k)[:, None]).to(torch.float32)
            idcg = (ideal / denom).sum(dim=1)
            zero = torch.zeros((), device=hit.device)
            out += [torch.where(valid, hit_num / k, zero).sum() / n_valid,
                    torch.where(valid, hit_num / lens.clamp(min=1), zero).sum() / n_valid,
                    torch.where(valid, dcg / idcg.clamp(min=1e-30), zero).sum() / n_valid]
        vals = torch.stack(out).cpu().numpy()
        for i, k in enumerate(self.topks):
            results['Precision'][k], results['Recall'][k], results['NDCG'][k] = (np.float32(v) for v in vals[3 * i:3 * i + 3])
        return results

    def _split_version(self, *splits):
        """What the device copies of a dataset's lists are keyed on: the dataset object and the version
        counters of the splits involved (bumped by list assignment and by dataset.invalidate())."""
        ver = getattr(self.dataset, 'version', None)
        return (id(self.dataset),) + tuple(ver(s) if ver else None for s in splits)

    def _eval_lists_device(self, val_or_test):
        """Device CSR of the evaluated lists (rowptr, sorted items, lengths); rebuilt only when the
        split's version changed (assignment to the split or to one of its users' entries bumps it by itself;
        only an edit INSIDE one user's list needs dataset.invalidate(split))."""
        key = ('eval', val_or_test) + self._split_version(val_or_test)
        if self._excl_cache.get('eval_key') != key or None in key:
            rp, cl = _sorted_csr_device([self.dataset.csr(val_or_test, sort=False)], self.dataset.n_items, self.device)
            self._excl_cache['eval_key'] = key
            self._excl_cache['eval_val'] = (rp, cl, (rp[1:] - rp[:-1]).contiguous())
        return self._excl_cache['eval_val']

    def calculate_metrics(self, eval_data, rec_items):
        """Same signature as trainer.py:109: eval_data list-of-lists, rec_items
        numpy [U, k].  The membership loop runs as one device kernel."""
        from .dataset import lists_to_csr
        rowptr, col = lists_to_csr(eval_data, sort=True)
        rp, cl = _csr_to_device(rowptr, col, self.device)
        rec = torch.from_numpy(np.ascontiguousarray(rec_items, dtype=np.int64)).to(self.device)
        hit = ops.hit_matrix(rec, rp, cl).cpu().numpy()
        # duplicates inside a user's eval list do not change membership, but len() counts them
        return self._metrics_from_hits(hit, np.array([len(x) for x in eval_data], dtype=np.int32))

    def _exclusion(self, val_or_test):
        """Device CSR of the items masked per user: train lists, plus val lists for
        'test' (trainer.py:149-159).  Cached; train / val lists are not mutated by
        the evaluation protocols."""
        if val_or_test == 'train':
            return None, None
        key = ('excl', val_or_test) + self._split_version('train', 'val')
        if self._excl_cache.get('excl_key_' + val_or_test) != key or None in key:
            lists = [self.dataset.csr('train', sort=False)]
            if val_or_test == 'test':
                lists.append(self.dataset.csr('val', sort=False))
            self._excl_cache['excl_key_' + val_or_test] = key
            self._excl_cache['excl_val_' + val_or_test] = _sorted_csr_device(lists, self.dataset.n_items, self.device)
        return self._excl_cache['excl_val_' + val_or_test]

    def recommend_all(self, val_or_test, banned_items=None, mode='auto'):
        """[n_users, max(topks)] recommended item ids on the device."""
        k = max(self.topks)
        excl_rowptr, excl_col = self._exclusion(val_or_test)
        banned = None
        if banned_items is not None:
            banned = torch.zeros(self.dataset.n_items, dtype=torch.uint8, device=self.device)
            banned[torch.as_tensor(np.asarray(banned_items), dtype=torch.int64, device=self.device)] = 1
        out = []
        with torch.no_grad():
            for start in range(0, self.dataset.n_users, self.eval_chunk):
                users = None if self.eval_chunk >= self.dataset.n_users else \
                    torch.arange(start, min(start + self.eval_chunk, self.dataset.n_users), dtype=torch.int64, device=self.device)
                out.append(self.model.recommend(users, k, excl_rowptr, excl_col, banned, **({} if mode == 'auto' else {'mode': mode})))
        return torch.cat(out, dim=0) if len(out) > 1 else out[0]

    def eval(self, val_or_test, banned_items=None, _eval_lists=None):
        """trainer.py:140-177; returns (results string, metrics dict).  _eval_lists (internal): device CSR
        (rowptr, sorted items, lengths) evaluated instead of the dataset's lists of that stage."""
        self.model.eval()
        rec = self.recommend_all(val_or_test, banned_items)
        rp, cl, lens = _eval_lists if _eval_lists is not None else self._eval_lists_device(val_or_test)
        if self.config.get('host_metrics', False):
            hit = ops.hit_matrix(rec.contiguous(), rp, cl)
            metrics = self._metrics_from_hits(hit.cpu().numpy(), lens.cpu().numpy().astype(np.int32))
        elif len(self.topks) <= MAX_METRIC_CUTS and rec.shape[0] > 0:
            metrics = self._metrics_device(rec, rp, cl)
        else:
            metrics = self._metrics_from_hits_device(ops.hit_matrix(rec.contiguous(), rp, cl), lens)
        self.last_rec_items = rec

        def row(name):
            return ''.join(f'{metrics[name][k] * 100.:.3f}%@{k:d}, ' for k in self.topks)
        results = f"Precision: {row('Precision')}Recall: {row('Recall')}NDCG: {row('NDCG')}"
        return results, metrics

    def inductive_eval(self, n_old_users, n_old_items):
        """The six test evaluations of the inductive protocol (trainer.py:179-219) on a model whose
        graph was extended by users >= n_old_users and items >= n_old_items: every combination the
        paper reports of {all, old, new} users x {all, old, new} items.  A user outside the evaluated
        user set gets an empty test list; for an item subset the test lists are filtered to it and
        the other items are banned from the recommendations.  The reference edits dataset.test_data in place
        and restores it; here the six filtered lists are cut from the test CSR on the device and the dataset is
        never touched."""
        n_users, n_items = self.dataset.n_users, self.dataset.n_items
        rp, cl, _ = self._eval_lists_device('test')
        user_of = torch.repeat_interleave(torch.arange(n_users, device=self.device), rp[1:] - rp[:-1])
        old_items = np.arange(n_old_items)
        new_items = np.arange(n_old_items, n_items)
        #  label                     evaluated users (lo, hi)    kept items  banned items
        variants = [
            ('All users and all items', (0, n_users),               None,  None),
            ('Old users and all items', (0, n_old_users),           None,  None),
            ('New users and all items', (n_old_users, n_users),     None,  None),
            ('All users and old items', (0, n_users),               'old', new_items),
            ('All users and new items', (0, n_users),               'new', old_items),
            ('Old users and old items', (0, n_old_users),           'old', new_items),
        ]
        for label, (ulo, uhi), keep, banned in variants:
            m = (user_of >= ulo) & (user_of < uhi)
            if keep == 'old':
                m &= cl < n_old_items
            elif keep == 'new':
                m &= cl >= n_old_items
            lens = torch.bincount(user_of[m], minlength=n_users)
            frp = torch.zeros(n_users + 1, dtype=torch.int64, device=self.device)
            torch.cumsum(lens, 0, out=frp[1:])
            summary, _ = self.eval('test', banned_items=banned, _eval_lists=(frp, cl[m].contiguous(), lens))
            print(f'{label} result. {summary}')


class BPRTrainer(BasicTrainer):
    """trainer.py:222-248."""

    def __init__(self, trainer_config):
        super().__init__(trainer_config)
        self.batch_size = trainer_config['batch_size']
        self.initialize_optimizer()
        self.l2_reg = trainer_config['l2_reg']
        self._graph = None

    @functools.cached_property
    def sampler(self):
        """Built on first use: a trainer made only to evaluate (the inductive scripts) never pays for it."""
        return DeviceSampler(self.dataset, self.device, self.config.get('seed', 2021))

    def bpr_step(self, inputs):
        """One optimisation step on an int64 [B, 3] batch; returns the loss tensor.  Full-size batches replay one
        captured HIP graph like node_step (MF has no node-id path), unless the loss needs a collective (a
        column-sharded model) or the config says 'hip_graph': False."""
        graph = self._graph_wanted() and getattr(self.model, 'slice_reduce_fn', None) is None
        self._seed_for_step(graph)
        if graph and inputs.shape[0] == self.batch_size:
            return self._graph_step((inputs,), self._triplet_loss, kind='triplets')
        return self._optimise_loss(self._triplet_loss(inputs))

    def _triplet_loss(self, inputs):
        users, pos_items, neg_items = inputs.t().contiguous().unbind(0)          # one transpose, three row views
        if hasattr(self.model, 'bpr_loss'):                                      # MF: the scalar loss as one autograd node
            return self.model.bpr_loss(users, pos_items, neg_items, self.l2_reg)
        terms = self.model.bpr_loss_terms(users, pos_items, neg_items)
        return terms[0] + self.l2_reg * terms[1]

    def _optimise_loss(self, loss):
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.detach()

    def node_step(self, nodes):
        """One optimisation step on the node ids of a batch (DeviceSampler.epoch_node_batches).  Full-size batches
        replay ONE captured HIP graph (forward, backward and the fused Adam step: 22 launches become one) unless the
        trainer config says 'hip_graph': False — what a launch-bound step needs (Gowalla-size LightGCN, IMF: -20...-25 %);
        a GPU-bound step pays ~1 % for the copies into the static buffers (profiles/r02n_*)."""
        loss_fn = lambda n: self.model.bpr_loss_nodes(n, self.l2_reg)
        graph = self._graph_wanted()
        self._seed_for_step(graph)
        if graph and nodes.numel() == 3 * self.batch_size:
            return self._graph_step((nodes,), loss_fn)
        return self._optimise_loss(loss_fn(nodes))

    def flat_step(self, flat):
        """One optimisation step of a model without propagation (MF) on a batch as int64 [3 B] = users | positives |
        negatives (DeviceSampler.epoch_node_batches with item_offset 0): the three id vectors are views — no transpose
        kernel — and full-size batches replay one captured HIP graph."""
        def loss_fn(f):
            b = f.numel() // 3
            return self.model.bpr_loss(f[:b], f[b:2 * b], f[2 * b:], self.l2_reg)
        graph = self._graph_wanted()
        self._seed_for_step(graph)
        if graph and flat.numel() == 3 * self.batch_size:
            return self._graph_step((flat,), loss_fn, kind='flat')
        return self._optimise_loss(loss_fn(flat))

    def train_one_epoch(self):
        losses = AverageMeter()
        pending = []
        if hasattr(self.model, 'bpr_loss') and getattr(self.model, 'slice_reduce_fn', None) is None \
                and not hasattr(self.model, 'bpr_loss_terms_nodes'):
            for flat in self.sampler.epoch_node_batches(self.batch_size, 0, into=self._draw_into(0, lambda b: (3 * b,))):
                pending.append((self.flat_step(flat), flat.shape[0] // 3))
        elif hasattr(self.model, 'bpr_loss_terms_nodes') and self.model.slice_reduce_fn is None:
            for nodes in self.sampler.epoch_node_batches(self.batch_size, self.model.n_users, into=self._draw_into(0, lambda b: (3 * b,))):
                pending.append((self.node_step(nodes), nodes.shape[0] // 3))
        else:
            for inputs in self.sampler.epoch_batches(self.batch_size, into=self._draw_into(0, lambda b: (b, 3))):
                pending.append((self.bpr_step(inputs), inputs.shape[0]))
        for loss, n in pending:               # one host sync per epoch, not per step (trainer.py:247)
            losses.update(loss.item(), n)
        return losses.avg


class IGCNTrainer(BasicTrainer):
    """trainer.py:281-320: BPR loss + auxiliary self-enhanced loss on the raw
    template embeddings; the feature matrix is annealed once per epoch."""

    def __init__(self, trainer_config):
        super().__init__(trainer_config)
        self.batch_size = trainer_config['batch_size']
        self.initialize_optimizer()
        self.l2_reg = trainer_config['l2_reg']
        self.aux_reg = trainer_config['aux_reg']
        self._graph = None

    # samplers and the re-indexed auxiliary dataset (dataset.py:258-273) are built on first use: the inductive
    # scripts make an IGCNTrainer only to evaluate (run/dropui/igcn_dropui.py:33-35)
    @functools.cached_property
    def sampler(self):
        return DeviceSampler(self.dataset, self.device, self.config.get('seed', 2021))

    @functools.cached_property
    def aux_dataset(self):
        return AuxiliaryDataset(self.dataset, self.model.user_map, self.model.item_map)

    @functools.cached_property
    def aux_sampler(self):
        return DeviceSampler(self.aux_dataset, self.device, self.config.get('seed', 2021) + 1)

    def igcn_step(self, inputs, aux_inputs):
        users, pos_items, neg_items = inputs.t().contiguous().unbind(0)
        return self._igcn_optimise(self.model.bpr_loss_terms(users, pos_items, neg_items), aux_inputs)

    def igcn_node_step(self, nodes, aux_inputs):
        """One step on node-id batches.  Full-size batches replay ONE captured HIP graph (both losses, backward, fused
        Adam) unless the config says 'hip_graph': False; the dropout seed then lives in device memory
        (IGCN.use_device_seed) and is advanced before every step, captured or not — the same sequence of seeds as
        the eager path draws."""
        graph = self._graph_wanted()
        self._seed_for_step(graph)
        if graph and nodes.numel() == 3 * self.batch_size and aux_inputs.shape[0] == self.batch_size:
            return self._graph_step((nodes, aux_inputs), self._igcn_loss)
        loss = self._igcn_loss(nodes, aux_inputs)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.detach()

    def _igcn_loss(self, nodes, aux_inputs):
        if self.config.get('fused_inmo_step', True) and hasattr(self.model, 'step_loss_nodes'):
            return self.model.step_loss_nodes(nodes, aux_inputs, self.l2_reg, self.aux_reg)       # one autograd node
        a_users, a_pos, a_neg = aux_inputs.t().contiguous().unbind(0)
        return self.model.bpr_loss_nodes(nodes, self.l2_reg) + self.aux_reg * self.model.aux_loss(a_users, a_pos, a_neg)

    def _igcn_optimise(self, terms, aux_inputs, main_loss=None):
        a_users, a_pos, a_neg = aux_inputs.t().contiguous().unbind(0)
        aux_loss = self.model.aux_loss(a_users, a_pos, a_neg)
        if main_loss is None:
            main_loss = terms[0] + self.l2_reg * terms[1]
        loss = main_loss + self.aux_reg * aux_loss
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.detach()

    def train_one_epoch(self):
        losses = AverageMeter()
        pending = []
        if self.model.slice_reduce_fn is None:
            for nodes, aux_inputs in zip(self.sampler.epoch_node_batches(self.batch_size, self.model.n_users,
                                                                         into=self._draw_into(0, lambda b: (3 * b,))),
                                         self.aux_sampler.epoch_batches(self.batch_size, into=self._draw_into(1, lambda b: (b, 3)))):
                pending.append((self.igcn_node_step(nodes, aux_inputs), nodes.shape[0] // 3))
        else:
            for inputs, aux_inputs in zip(self.sampler.epoch_batches(self.batch_size),
                                          self.aux_sampler.epoch_batches(self.batch_size)):
                pending.append((self.igcn_step(inputs, aux_inputs), inputs.shape[0]))
        for loss, n in pending:
            losses.update(loss.item(), n)
        self.model.feat_mat_anneal()
        return losses.avg

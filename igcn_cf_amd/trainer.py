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
import os
import sys
import time

import numpy as np
import torch
from torch.optim import SGD, Adam  # noqa: F401  (resolved by name, trainer.py:43-45)

from . import ops
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


def _csr_to_device(rowptr, col, device):
    return (torch.from_numpy(np.ascontiguousarray(rowptr, dtype=np.int64)).to(device),
            torch.from_numpy(np.ascontiguousarray(col, dtype=np.int32)).to(device))


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
        rowptr, col = dataset.csr('train', sort=True)
        self.rowptr, self.col = _csr_to_device(rowptr, col, device)
        nonempty = np.flatnonzero(np.diff(rowptr) > 0).astype(np.int32)
        self.nonempty = torch.from_numpy(nonempty).to(device)
        self.n_items = dataset.n_items
        self.length = len(dataset)
        self.seed = int(seed)
        self.calls = 0

    def epoch_batches(self, batch_size):
        """ceil(len / batch_size) batches, the last one short, as a DataLoader over
        a dataset of len(train_array) draws (dataset.py:116-117)."""
        left = self.length
        while left > 0:
            b = min(batch_size, left)
            left -= b
            self.calls += 1
            yield ops.bpr_sample(self.rowptr, self.col, self.nonempty, self.n_items, b,
                                 (self.seed * 0x9E3779B97F4A7C15 + self.calls) & 0xFFFFFFFFFFFFFFFF)


class BasicTrainer:
    def __init__(self, trainer_config):
        self.config = trainer_config
        self.name = trainer_config['name']
        self.dataset = trainer_config['dataset']
        self.model = trainer_config['model']
        self.topks = trainer_config['topks']
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
        self.opt = opt(self.model.parameters(), lr=self.config['lr'], **kw)

    def train_one_epoch(self):
        raise NotImplementedError

    def record(self, writer, stage, metrics):
        for metric in metrics:
            for k in self.topks:
                writer.add_scalar('{:s}_{:s}/{:s}_{:s}@{:d}'.format(self.model.name, self.name, stage, metric, k),
                                  metrics[metric][k], self.epoch)

    def train(self, verbose=True, writer=None):
        """Epoch loop with validation, best-NDCG checkpoint and patience (trainer.py:57-107)."""
        if not self.model.trainable:
            results, metrics = self.eval('val')
            if verbose:
                print('Validation result. {:s}'.format(results))
            return metrics['NDCG'][self.topks[0]]

        if not os.path.exists('checkpoints'):
            os.mkdir('checkpoints')
        patience = self.max_patience
        for self.epoch in range(self.n_epochs):
            start_time = time.time()
            self.model.train()
            loss = self.train_one_epoch()
            _, metrics = self.eval('train')
            consumed_time = time.time() - start_time
            if verbose:
                print('Epoch {:d}/{:d}, Loss: {:.6f}, Time: {:.3f}s'.format(self.epoch, self.n_epochs, loss, consumed_time))
            if writer:
                writer.add_scalar('{:s}_{:s}/train_loss'.format(self.model.name, self.name), loss, self.epoch)
                self.record(writer, 'train', metrics)
            if (self.epoch + 1) % self.val_interval != 0:
                continue

            start_time = time.time()
            results, metrics = self.eval('val')
            consumed_time = time.time() - start_time
            if verbose:
                print('Validation result. {:s}Time: {:.3f}s'.format(results, consumed_time))
            if writer:
                self.record(writer, 'validation', metrics)

            ndcg = metrics['NDCG'][self.topks[0]]
            if ndcg > self.best_ndcg:
                if self.save_path:
                    os.remove(self.save_path)
                self.save_path = os.path.join('checkpoints', '{:s}_{:s}_{:s}_{:.3f}.pth'
                                              .format(self.model.name, self.name, self.dataset.name, ndcg * 100))
                self.best_ndcg = ndcg
                self.model.save(self.save_path)
                patience = self.max_patience
                print('Best NDCG, save model to {:s}'.format(self.save_path))
            else:
                patience -= self.val_interval
                if patience <= 0:
                    print('Early stopping!')
                    break
        self.model.load(self.save_path)
        return self.best_ndcg

    # ---- metrics ---------------------------------------------------------------
    def _metrics_from_hits(self, hit_matrix, eval_data_len):
        """trainer.py:116-137 verbatim in numpy dtypes (float32 hits, int32 lengths)."""
        results = {'Precision': {}, 'Recall': {}, 'NDCG': {}}
        for k in self.topks:
            hit_num = np.sum(hit_matrix[:, :k], axis=1)
            precisions = hit_num / k
            with np.errstate(invalid='ignore', divide='ignore'):
                recalls = hit_num / eval_data_len
            max_hit_num = np.minimum(eval_data_len, k)
            max_hit_matrix = (np.arange(k)[None, :] < max_hit_num[:, None]).astype(np.float32)
            denominator = np.log2(np.arange(2, k + 2, dtype=np.float32))[None, :]
            dcgs = np.sum(hit_matrix[:, :k] / denominator, axis=1)
            idcgs = np.sum(max_hit_matrix / denominator, axis=1)
            with np.errstate(invalid='ignore', divide='ignore'):
                ndcgs = dcgs / idcgs
            user_masks = (max_hit_num > 0)
            results['Precision'][k] = precisions[user_masks].mean()
            results['Recall'][k] = recalls[user_masks].mean()
            results['NDCG'][k] = ndcgs[user_masks].mean()
        return results

    def _metrics_from_hits_device(self, hit, eval_len):
        """The reductions of trainer.py:116-137 on the device (float32, same formulas): only the
        final scalars cross PCIe.  Used by eval(); calculate_metrics() keeps the numpy path, whose
        results are bit-identical to the reference's."""
        results = {'Precision': {}, 'Recall': {}, 'NDCG': {}}
        lens = eval_len.to(torch.float32)
        valid = eval_len > 0
        n_valid = valid.sum().clamp(min=1).to(torch.float32)
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

    def _eval_lists_device(self, val_or_test):
        """Device CSR of the evaluated lists; rebuilt whenever the dataset hands out new arrays
        (dataset.invalidate() after the lists were edited in place)."""
        if hasattr(self.dataset, 'invalidate'):
            self.dataset.invalidate(val_or_test)
        rowptr, col = self.dataset.csr(val_or_test, sort=True)
        key = ('eval', val_or_test, id(rowptr), id(col))
        if self._excl_cache.get('eval_key') != key:
            rp, cl = _csr_to_device(rowptr, col, self.device)
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
        key = (val_or_test, id(self.dataset))
        if key not in self._excl_cache:
            excl = self.dataset.csr('train', sort=True)
            if val_or_test == 'test':
                excl = _merge_sorted_csr(excl, self.dataset.csr('val', sort=True))
            self._excl_cache[key] = _csr_to_device(excl[0], excl[1], self.device)
        return self._excl_cache[key]

    def recommend_all(self, val_or_test, banned_items=None):
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
                users = torch.arange(start, min(start + self.eval_chunk, self.dataset.n_users), dtype=torch.int64,
                                     device=self.device)
                out.append(self.model.recommend(users, k, excl_rowptr, excl_col, banned))
        return torch.cat(out, dim=0) if len(out) > 1 else out[0]

    def eval(self, val_or_test, banned_items=None):
        """trainer.py:140-177; returns (results string, metrics dict)."""
        self.model.eval()
        rec = self.recommend_all(val_or_test, banned_items)
        rp, cl, lens = self._eval_lists_device(val_or_test)    # eval lists may have been edited in place
        hit = ops.hit_matrix(rec.contiguous(), rp, cl)
        if self.config.get('host_metrics', False):
            metrics = self._metrics_from_hits(hit.cpu().numpy(), lens.cpu().numpy().astype(np.int32))
        else:
            metrics = self._metrics_from_hits_device(hit, lens)
        self.last_rec_items = rec

        precison = ''
        recall = ''
        ndcg = ''
        for k in self.topks:
            precison += '{:.3f}%@{:d}, '.format(metrics['Precision'][k] * 100., k)
            recall += '{:.3f}%@{:d}, '.format(metrics['Recall'][k] * 100., k)
            ndcg += '{:.3f}%@{:d}, '.format(metrics['NDCG'][k] * 100., k)
        results = 'Precision: {:s}Recall: {:s}NDCG: {:s}'.format(precison, recall, ndcg)
        return results, metrics

    def inductive_eval(self, n_old_users, n_old_items):
        """The six masked evaluations of trainer.py:179-219."""
        test_data = self.dataset.test_data.copy()

        def restore():
            self.dataset.test_data = test_data.copy()

        results, _ = self.eval('test')
        print('All users and all items result. {:s}'.format(results))

        for user in range(n_old_users, self.dataset.n_users):
            self.dataset.test_data[user] = []
        results, _ = self.eval('test')
        print('Old users and all items result. {:s}'.format(results))

        restore()
        for user in range(n_old_users):
            self.dataset.test_data[user] = []
        results, _ = self.eval('test')
        print('New users and all items result. {:s}'.format(results))

        restore()
        for user in range(self.dataset.n_users):
            test_items = np.array(self.dataset.test_data[user])
            self.dataset.test_data[user] = test_items[test_items < n_old_items].tolist()
        results, _ = self.eval('test', banned_items=np.arange(n_old_items, self.dataset.n_items))
        print('All users and old items result. {:s}'.format(results))

        restore()
        for user in range(self.dataset.n_users):
            test_items = np.array(self.dataset.test_data[user])
            self.dataset.test_data[user] = test_items[test_items >= n_old_items].tolist()
        results, _ = self.eval('test', banned_items=np.arange(n_old_items))
        print('All users and new items result. {:s}'.format(results))

        restore()
        for user in range(n_old_users, self.dataset.n_users):
            self.dataset.test_data[user] = []
        for user in range(n_old_users):
            test_items = np.array(self.dataset.test_data[user])
            self.dataset.test_data[user] = test_items[test_items < n_old_items].tolist()
        results, _ = self.eval('test', banned_items=np.arange(n_old_items, self.dataset.n_items))
        print('Old users and old items result. {:s}'.format(results))

        restore()


class BPRTrainer(BasicTrainer):
    """trainer.py:222-248."""

    def __init__(self, trainer_config):
        super().__init__(trainer_config)
        self.batch_size = trainer_config['batch_size']
        self.sampler = DeviceSampler(self.dataset, self.device, trainer_config.get('seed', 2021))
        self.initialize_optimizer()
        self.l2_reg = trainer_config['l2_reg']

    def bpr_step(self, inputs):
        """One optimisation step on an int64 [B, 3] batch; returns the loss tensor."""
        users, pos_items, neg_items = inputs.t().contiguous().unbind(0)          # one transpose, three row views
        terms = self.model.bpr_loss_terms(users, pos_items, neg_items)
        loss = terms[0] + self.l2_reg * terms[1]
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.detach()

    def train_one_epoch(self):
        losses = AverageMeter()
        pending = []
        for inputs in self.sampler.epoch_batches(self.batch_size):
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
        seed = trainer_config.get('seed', 2021)
        self.sampler = DeviceSampler(self.dataset, self.device, seed)
        self.aux_dataset = AuxiliaryDataset(self.dataset, self.model.user_map, self.model.item_map)
        self.aux_sampler = DeviceSampler(self.aux_dataset, self.device, seed + 1)
        self.initialize_optimizer()
        self.l2_reg = trainer_config['l2_reg']
        self.aux_reg = trainer_config['aux_reg']

    def igcn_step(self, inputs, aux_inputs):
        users, pos_items, neg_items = inputs.t().contiguous().unbind(0)
        terms = self.model.bpr_loss_terms(users, pos_items, neg_items)
        a_users, a_pos, a_neg = aux_inputs.t().contiguous().unbind(0)
        aux_loss = self.model.aux_loss(a_users, a_pos, a_neg)
        loss = terms[0] + self.l2_reg * terms[1] + self.aux_reg * aux_loss
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.detach()

    def train_one_epoch(self):
        losses = AverageMeter()
        pending = []
        for inputs, aux_inputs in zip(self.sampler.epoch_batches(self.batch_size),
                                      self.aux_sampler.epoch_batches(self.batch_size)):
            pending.append((self.igcn_step(inputs, aux_inputs), inputs.shape[0]))
        for loss, n in pending:
            losses.update(loss.item(), n)
        self.model.feat_mat_anneal()
        return losses.avg

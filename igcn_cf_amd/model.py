"""MF / LightGCN / IGCN (INMO-LightGCN) / IMF (INMO-MF) on the HIP kernels.

Host-side mirror of the reference's model contract (model.py:16-21 get_model,
:31-49 BasicModel, :52-72 MF, :75-123 LightGCN, :354-466 IGCN, :536-543 IMF):
same class names, constructor config keys, attributes, method names, argument
meaning and checkpoint keys, so the reference's run scripts and trainers read
the same against this package.  What differs is what runs underneath:

* the normalised adjacency / template feature matrix are CSR in HBM
  (graph.CsrMatrix) built once, not a COO tensor turned into a DGL graph on
  every call (model.py:99-100, :428-429, :439-440);
* get_rep() is K launches of the hand-written SpMM with the layer mean fused
  into the last one; in eval mode the result is cached until the parameters or
  the graph change (the reference recomputes it for every 512-user batch,
  model.py:119 — identical results, dropout is off in eval, model.py:264-265);
* bpr_loss_terms() / recommend() are the fused kernels the trainers use;
  bpr_forward() / predict() keep the reference's signatures for callers that
  want gathered rows or a dense score block.

Baseline models outside the hot path (NGCF, IDCF, IMCGAE, MultiVAE, NeuMF,
ItemKNN, Popularity) are out of scope.
"""
import sys

import numpy as np
import torch
import torch.nn as nn
from torch.nn.init import normal_

from . import _lib, ops
from .graph import (feature_matrix_device, graph_rank_nodes, normalized_adjacency_device)


def get_model(config, dataset):
    """Factory by class name (model.py:16-21)."""
    config = config.copy()
    config['dataset'] = dataset
    cls = getattr(sys.modules[__name__], config['name'])
    return cls(config)


def _sq_norms(*row_sets):
    """Sum over the given [B, d] tensors of their per-row squared L2 norms -> [B]."""
    return sum(r.square().sum(dim=1) for r in row_sets)


class BasicModel(nn.Module):
    def __init__(self, model_config):
        super().__init__()
        self.config = model_config
        self.name = model_config['name']
        self.device = torch.device(model_config['device'])
        if self.device.type != 'cuda':
            raise _lib.IgcnError('%s runs on the MI355X HIP kernels only; device must be cuda' % self.name)
        self.n_users = model_config['dataset'].n_users
        self.n_items = model_config['dataset'].n_items
        self.trainable = True
        self._rep_cache = None
        # set by dist.column_shard_model: this model holds d/P embedding columns and the partial dots of a
        # loss are summed over the ranks with this function (an all-reduce)
        self.slice_reduce_fn = None
        self._batch_grads = ops.BatchGradTable()

    def predict(self, users):
        raise NotImplementedError

    def train(self, mode=True):
        # Entering or leaving training mode drops the cached eval-mode representation: fused optimizers
        # update parameters without bumping tensor version counters, so the version in the cache key
        # alone cannot be trusted across training steps.  eval() -> eval() keeps the cache (the
        # reference evaluates 'train' and 'val' back to back every epoch, trainer.py:71, :84).
        if mode or self.training:
            self._rep_cache = None
        return super().train(mode)

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path, map_location=self.device))

    # ---- fused entry points used by the trainers -------------------------------
    def score_tables(self):
        """(user_rows [>=n_users, d], item_rows [n_items, d]) for scoring."""
        rep = self.get_rep()
        return rep, rep[self.n_users:]

    def recommend(self, users, k, excl_rowptr=None, excl_col=None, banned=None, mode='auto'):
        """Top-k item ids per user (best first), masked by the exclusion CSR and the
        banned mask: predict -> mask -> topk of trainer.py:147-163, fused.  users: int64 ids on the GPU, or None = all users in
        order.  mode: ops.score_topk's ('exact' = the fp32 sweep alone; the lists are the same either way)."""
        user_rows, item_rows = self.score_tables()
        if users is None:                  # every user, in order: no id list to follow, no exclusion rows to pick out
            idx, _ = ops.score_topk(user_rows, item_rows, k, batch=self.n_users, excl_rowptr=excl_rowptr, excl_col=excl_col,
                                    banned=banned, mode=mode)
            return idx
        idx, _ = ops.score_topk(user_rows, item_rows, k, user_ids=users.contiguous(), excl_rowptr=excl_rowptr,
                                excl_col=excl_col, banned=banned, mode=mode)
        return idx

    def _cached_rep(self, key, compute):
        if self.training or torch.is_grad_enabled():
            return compute()
        if self._rep_cache is None or self._rep_cache[0] != key:
            self._rep_cache = (key, compute())
        return self._rep_cache[1]


class MF(BasicModel):
    """model.py:52-72."""

    def __init__(self, model_config):
        super().__init__(model_config)
        self.embedding_size = model_config['embedding_size']
        self.user_embedding = nn.Embedding(self.n_users, self.embedding_size)
        self.item_embedding = nn.Embedding(self.n_items, self.embedding_size)
        normal_(self.user_embedding.weight, std=0.1)
        normal_(self.item_embedding.weight, std=0.1)
        self.to(device=self.device)

    def bpr_forward(self, users, pos_items, neg_items):
        """Reference signature (model.py:62-67): gathered rows + per-triplet squared L2 norm."""
        rows = (self.user_embedding(users), self.item_embedding(pos_items), self.item_embedding(neg_items))
        return (*rows, _sq_norms(*rows))

    def bpr_loss_terms(self, users, pos_items, neg_items):
        u, i = self.user_embedding.weight, self.item_embedding.weight
        return ops.bpr_loss_terms(u, i, u, i, None, users, pos_items, neg_items, reduce_fn=self.slice_reduce_fn)

    def bpr_loss(self, users, pos_items, neg_items, l2_reg):
        """The scalar training loss of trainer.py:242 as one autograd node (column-sharded slices keep the two-term path)."""
        if self.slice_reduce_fn is not None:
            terms = self.bpr_loss_terms(users, pos_items, neg_items)
            return terms[0] + l2_reg * terms[1]
        return ops.bpr_scalar_loss(self.user_embedding.weight, self.item_embedding.weight, users, pos_items, neg_items, l2_reg)

    def score_tables(self):
        return self.user_embedding.weight.detach(), self.item_embedding.weight.detach()

    def predict(self, users):
        user_e = self.user_embedding(users)
        return torch.mm(user_e, self.item_embedding.weight.t())


class LightGCN(BasicModel):
    """model.py:75-123."""

    def __init__(self, model_config):
        super().__init__(model_config)
        self.embedding_size = model_config['embedding_size']
        self.n_layers = model_config['n_layers']
        self.embedding = nn.Embedding(self.n_users + self.n_items, self.embedding_size)
        self.norm_adj = self.generate_graph(model_config['dataset'])
        normal_(self.embedding.weight, std=0.1)
        self.to(device=self.device)

    def generate_graph(self, dataset):
        """A_hat = D^-1/2 A D^-1/2 as device CSR (model.py:85-94), built in HBM."""
        return normalized_adjacency_device(dataset.train_array, dataset.n_users, dataset.n_items, self.device)

    def get_rep(self, needed_rows=None):
        """needed_rows: ids of the only rows the caller will read (training batches); the
        propagation is then pruned to what those rows depend on (ops.propagate_mean)."""
        w = self.embedding.weight
        if needed_rows is not None and self.training:
            return ops.PropagateFn.apply(w, self.norm_adj, self.norm_adj, self.n_layers, needed_rows)
        key = (w._version, id(self.norm_adj), w.data_ptr())
        # A_hat is symmetric, so the same CSR serves the backward pass
        return self._cached_rep(key, lambda: ops.PropagateFn.apply(w, self.norm_adj, self.norm_adj, self.n_layers))

    def _batch_rows(self, users, pos_items, neg_items):
        if not self.config.get('prune_propagation', True):
            return None
        return torch.cat([users, self.n_users + pos_items, self.n_users + neg_items])

    def bpr_forward(self, users, pos_items, neg_items):
        """Reference signature (model.py:108-116): propagated rows of the triplets; the L2 term is
        taken on the RAW embedding rows."""
        rep, raw = self.get_rep(), self.embedding.weight
        idx = (users, self.n_users + pos_items, self.n_users + neg_items)
        return (*(rep[i] for i in idx), _sq_norms(*(raw[i] for i in idx)))

    def bpr_loss_terms(self, users, pos_items, neg_items):
        if self.slice_reduce_fn is not None:             # embedding-column slice: partial dots + all-reduce
            rep, e = self.get_rep(self._batch_rows(users, pos_items, neg_items)), self.embedding.weight
            return ops.bpr_loss_terms(rep, rep, e, e, None, users, pos_items, neg_items, self.n_users, self.n_users,
                                      reduce_fn=self.slice_reduce_fn)
        return self.bpr_loss_terms_nodes(torch.cat([users, self.n_users + pos_items, self.n_users + neg_items]))

    def bpr_loss_terms_nodes(self, nodes):
        """The same from the node ids of the batch, int64 [3 B] = users | n_users + positives | n_users + negatives
        (what ops.bpr_sample_nodes draws): one fused autograd node, row-sparse gradients."""
        return ops.graph_bpr_terms(self.embedding.weight, self.norm_adj, self.norm_adj, self.n_layers, nodes, 'raw',
                                   self._batch_grads, self.config.get('prune_propagation', True))

    def bpr_loss_nodes(self, nodes, l2_reg):
        """The scalar training loss of trainer.py:242 (bpr + l2_reg * mean l2_norm_sq) as one differentiable tensor."""
        return ops.graph_bpr_terms(self.embedding.weight, self.norm_adj, self.norm_adj, self.n_layers, nodes, 'raw',
                                   self._batch_grads, self.config.get('prune_propagation', True), l2_reg)

    def predict(self, users):
        rep = self.get_rep()
        return torch.mm(rep[users, :], rep[self.n_users:, :].t())


class IGCN(BasicModel):
    """INMO-LightGCN, model.py:354-466."""

    def __init__(self, model_config):
        super().__init__(model_config)
        self.embedding_size = model_config['embedding_size']
        self.n_layers = model_config['n_layers']
        self.dropout = model_config['dropout']
        self.feature_ratio = model_config['feature_ratio']
        self.norm_adj = self.generate_graph(model_config['dataset'])
        self.alpha = 1.
        self.delta = model_config.get('delta', 0.99)
        self.feat_mat, self.user_map, self.item_map, self.row_sum = \
            self.generate_feat(model_config['dataset'], ranking_metric=model_config.get('ranking_metric', 'sort'))
        self.update_feat_mat()

        self.embedding = nn.Embedding(self.feat_mat.shape[1], self.embedding_size)
        self.w = nn.Parameter(torch.ones([self.embedding_size], dtype=torch.float32, device=self.device))
        normal_(self.embedding.weight, std=0.1)
        self.to(device=self.device)
        self._drop_calls = 0
        self._seed_dev = None

    # feat_mat may be re-assigned on a live model (run/dropui/igcn_dropui.py:28-32)
    @property
    def feat_mat(self):
        return self._feat_mat

    @feat_mat.setter
    def feat_mat(self, value):
        self._feat_mat = value
        self._feat_scale = None
        self._rep_cache = None

    def update_feat_mat(self):
        """Edge value = row_sum[row]^((alpha-1)/2 - 0.5) (model.py:374-377), kept as
        one scale per ROW — the kernel applies it in its epilogue, the matrix
        structure is never rebuilt."""
        n_rows = self.feat_mat.shape[0]
        scale = getattr(self, '_feat_scale', None)
        if scale is None or scale.numel() != n_rows:       # else rewritten in place: a captured HIP graph keeps reading it
            scale = torch.empty(n_rows, dtype=torch.float32, device=self.device)
        expo = (self.alpha - 1.) / 2. - 0.5
        _lib.check(_lib.lib().igcn_csr_row_pow_f32(self.feat_mat.rowptr.data_ptr(), self.row_sum.data_ptr(), expo,
                                                   None, scale.data_ptr(), n_rows, _lib.current_stream()),
                   'igcn_csr_row_pow_f32')
        self._feat_scale = scale
        self._rep_cache = None

    def feat_mat_anneal(self):
        self.alpha *= self.delta
        self.update_feat_mat()

    def feat_values(self):
        """Explicit per-edge values of the feature matrix (what the reference stores)."""
        val = torch.empty(self.feat_mat.nnz, dtype=torch.float32, device=self.device)
        expo = (self.alpha - 1.) / 2. - 0.5
        _lib.check(_lib.lib().igcn_csr_row_pow_f32(self.feat_mat.rowptr.data_ptr(), self.row_sum.data_ptr(), expo,
                                                   val.data_ptr(), None, self.feat_mat.shape[0],
                                                   _lib.current_stream()), 'igcn_csr_row_pow_f32')
        return val

    def generate_graph(self, dataset):
        return LightGCN.generate_graph(self, dataset)

    def generate_feat(self, dataset, is_updating=False, ranking_metric=None):
        """Template feature matrix (model.py:386-421); returns
        (feat CsrMatrix, user_map, item_map, row_sum tensor)."""
        if not is_updating:
            if self.feature_ratio < 1.:
                ranked_users, ranked_items = graph_rank_nodes(dataset, ranking_metric)
                core_users = ranked_users[:int(self.n_users * self.feature_ratio)]
                core_items = ranked_items[:int(self.n_items * self.feature_ratio)]
            else:
                core_users = np.arange(self.n_users, dtype=np.int64)
                core_items = np.arange(self.n_items, dtype=np.int64)
            user_map = {int(u): idx for idx, u in enumerate(core_users)}
            item_map = {int(i): idx for idx, i in enumerate(core_items)}
        else:
            user_map, item_map = self.user_map, self.item_map
        full = not is_updating and self.feature_ratio >= 1.      # every node its own template: identity maps
        feat, row_sum = feature_matrix_device(dataset.train_array, self.n_users, self.n_items,
                                              None if full else user_map, None if full else item_map, self.device)
        return feat, user_map, item_map, row_sum

    def inductive_rep_layer(self, feat_mat, keep_prob=1., seed=0):
        """X0 = dropout(F) @ T (model.py:423-432; no padding tensor is needed)."""
        if self._feat_scale is None:
            self.update_feat_mat()
        return ops.FeatureLayerFn.apply(self.embedding.weight, feat_mat, feat_mat.transposed_view(),
                                        self._feat_scale, keep_prob, seed)

    def _dropout_args(self):
        """NGCF.dropout_sp_mat (model.py:263-275): train mode only; the mask is a
        hash of (seed, edge id) drawn per call from torch's CPU generator."""
        if not self.training or self.dropout <= 0.:
            return 1., 0
        if self._seed_dev is not None:                      # set by advance_dropout_seed(), read by the kernels
            return 1. - self.dropout, self._seed_dev
        return 1. - self.dropout, int(torch.randint(0, 2 ** 62, (1,)).item())

    def use_device_seed(self):
        """Keep the dropout seed in device memory (a step captured in a HIP graph must not bake it in as a launch
        argument); the trainer then calls advance_dropout_seed() once before every step."""
        if self._seed_dev is None:
            self._seed_dev = torch.zeros(1, dtype=torch.int64, device=self.device)

    def advance_dropout_seed(self):
        """The next seed of the same CPU-generator sequence _dropout_args draws from, written to the device."""
        if self._seed_dev is not None and self.training and self.dropout > 0.:
            self._seed_dev.fill_(int(torch.randint(0, 2 ** 62, (1,)).item()))

    def _compute_rep(self, needed_rows=None):
        keep_prob, seed = self._dropout_args()
        x0 = self.inductive_rep_layer(self.feat_mat, keep_prob, seed)
        return ops.PropagateFn.apply(x0, self.norm_adj, self.norm_adj, self.n_layers, needed_rows)

    def get_rep(self, needed_rows=None):
        if needed_rows is not None and self.training:
            return self._compute_rep(needed_rows)
        w = self.embedding.weight
        key = (w._version, id(self.norm_adj), id(self.feat_mat), w.data_ptr())     # anneal / feat_mat swap drop the cache themselves
        return self._cached_rep(key, self._compute_rep)

    _batch_rows = LightGCN._batch_rows

    def bpr_forward(self, users, pos_items, neg_items):
        """Reference signature (model.py:293-299 via :448-449): propagated rows; the L2 term is taken
        on those same propagated rows."""
        rep = self.get_rep()
        rows = tuple(rep[i] for i in (users, self.n_users + pos_items, self.n_users + neg_items))
        return (*rows, _sq_norms(*rows))

    def bpr_loss_terms(self, users, pos_items, neg_items):
        if self.slice_reduce_fn is not None:
            rep = self.get_rep(self._batch_rows(users, pos_items, neg_items))
            return ops.bpr_loss_terms(rep, rep, rep, rep, None, users, pos_items, neg_items, self.n_users, self.n_users,
                                      reduce_fn=self.slice_reduce_fn)
        return self.bpr_loss_terms_nodes(torch.cat([users, self.n_users + pos_items, self.n_users + neg_items]))

    def bpr_loss_terms_nodes(self, nodes):
        """As LightGCN.bpr_loss_terms_nodes; the L2 term is taken on the propagated rows (model.py:297-298)."""
        keep_prob, seed = self._dropout_args()
        x0 = self.inductive_rep_layer(self.feat_mat, keep_prob, seed)
        return ops.graph_bpr_terms(x0, self.norm_adj, self.norm_adj, self._prop_layers(), nodes, 'rep', self._batch_grads,
                                   self.config.get('prune_propagation', True))

    def bpr_loss_nodes(self, nodes, l2_reg):
        keep_prob, seed = self._dropout_args()
        x0 = self.inductive_rep_layer(self.feat_mat, keep_prob, seed)
        return ops.graph_bpr_terms(x0, self.norm_adj, self.norm_adj, self._prop_layers(), nodes, 'rep', self._batch_grads,
                                   self.config.get('prune_propagation', True), l2_reg)

    def step_loss_nodes(self, nodes, aux_inputs, l2_reg, aux_reg):
        """The whole training loss of IGCNTrainer (trainer.py:300-312: bpr + l2_reg * mean l2_norm_sq + aux_reg * auxiliary
        loss) as ONE differentiable scalar and one autograd node (ops.InmoStepFn).  nodes: int64 [3 B] node ids of the
        batch; aux_inputs: int64 [Ba, 3] triplets of the auxiliary dataset (template space)."""
        if self._feat_scale is None:
            self.update_feat_mat()
        keep_prob, seed = self._dropout_args()
        return ops.inmo_step_loss(self.embedding.weight, self.w, self.feat_mat, self.feat_mat.transposed_view(), self._feat_scale,
                                  keep_prob, seed, self.norm_adj, self._prop_layers(), nodes, aux_inputs, len(self.user_map),
                                  self._batch_grads, self.config.get('prune_propagation', True), l2_reg, aux_reg)

    def _prop_layers(self):
        return self.n_layers

    def aux_loss(self, users, pos_items, neg_items):
        """Self-enhanced auxiliary BPR loss on the raw template rows, weighted by w
        (trainer.py:304-311)."""
        e = self.embedding.weight
        off = len(self.user_map)
        return ops.bpr_loss_terms(e, e, None, None, self.w, users, pos_items, neg_items, off, 0,
                                  reduce_fn=self.slice_reduce_fn)[0]

    def predict(self, users):
        return LightGCN.predict(self, users)

    def save(self, path):
        params = {'sate_dict': self.state_dict(), 'user_map': self.user_map,
                  'item_map': self.item_map, 'alpha': self.alpha}
        torch.save(params, path)

    def load(self, path):
        params = torch.load(path, map_location=self.device, weights_only=False)
        self.load_state_dict(params['sate_dict'])
        self.user_map = params['user_map']
        self.item_map = params['item_map']
        self.alpha = params['alpha']
        self.feat_mat, _, _, self.row_sum = self.generate_feat(self.config['dataset'], is_updating=True)
        self.update_feat_mat()


class IMF(IGCN):
    """INMO-MF, model.py:536-543: the template layer without propagation."""

    def _prop_layers(self):
        return 0

    def _compute_rep(self, needed_rows=None):
        keep_prob, seed = self._dropout_args()
        return self.inductive_rep_layer(self.feat_mat, keep_prob, seed)

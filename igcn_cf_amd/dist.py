"""Row-sharded propagation across the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Not in the reference (single process, single device — SURVEY.md section 2.1); this is
the scheme BASELINE.json's north_star asks for at Amazon-book scale and above:

* the user set and the item set are each cut into P equal contiguous blocks
  (padded); rank r owns user block r and item block r: those rows of A_hat as
  local CSR (global, padded column ids), of the embeddings and of the outputs;
* every layer input X_l is replicated (the all-gather target); a rank computes
  its own rows of X_{l+1} straight into its slot of the next replicated buffer
  and the slots are exchanged with an in-place all-gather;
* the graph is bipartite: user rows read only item embeddings and vice versa.
  Each layer is therefore two half-steps, and the order of the halves alternates
  from layer to layer (users,items / items,users / ...), so that every
  all-gather runs under the SpMM of the other half — no half-step ever waits
  for a collective that was issued immediately before it;
* the layer mean is the epilogue of the last local SpMM, on owned rows only.

`spmm_fn` is the local product; the product path uses the HIP kernel
(ops.spmm).  The CPU tests inject a checker implementation to exercise the
partitioning / exchange logic under gloo with world_size 2.
"""
import numpy as np
import torch
import torch.distributed as dist

from .graph import normalized_adjacency_host


def block_size(n, world):
    return (n + world - 1) // world


class ShardLayout:
    """Padded global layout [P*bu user rows ; P*bi item rows]."""

    def __init__(self, n_users, n_items, world):
        self.n_users, self.n_items, self.world = n_users, n_items, world
        self.bu, self.bi = block_size(n_users, world), block_size(n_items, world)
        self.pu, self.pi = self.bu * world, self.bi * world          # padded section sizes
        self.n_pad = self.pu + self.pi

    def pad_index(self, node):
        """global node id (users first, then items) -> row in the padded layout."""
        node = np.asarray(node, dtype=np.int64)
        return np.where(node < self.n_users, node, node - self.n_users + self.pu)

    def user_rows(self, rank):
        lo = rank * self.bu
        return lo, min(lo + self.bu, self.n_users)

    def item_rows(self, rank):
        lo = rank * self.bi
        return lo, min(lo + self.bi, self.n_items)


def local_blocks_host(rowptr, col, val, layout, rank):
    """Rows of the global CSR owned by `rank`, as two CSR blocks (user rows,
    item rows) of exactly bu / bi rows (padding rows empty), padded column ids."""
    out = []
    for (lo, hi), nb, base in ((layout.user_rows(rank), layout.bu, 0),
                               (layout.item_rows(rank), layout.bi, layout.n_users)):
        s, e = rowptr[base + lo], rowptr[base + max(hi, lo)]
        rp = np.full(nb + 1, e - s, dtype=np.int64)
        rp[:max(hi - lo, 0) + 1] = rowptr[base + lo: base + max(hi, lo) + 1] - s
        out.append((rp, layout.pad_index(col[s:e]).astype(np.int32), val[s:e].copy()))
    return out


class RowShardedPropagator:
    def __init__(self, train_array, n_users, n_items, n_layers, rank, world, device, group=None,
                 spmm_fn=None, csr_factory=None, adjacency=None):
        self.layout = ShardLayout(n_users, n_items, world)
        self.n_layers, self.rank, self.world, self.group = n_layers, rank, world, group
        self.device = torch.device(device)
        if spmm_fn is None:
            from . import ops
            spmm_fn = ops.spmm
        if csr_factory is None:
            from .graph import CsrMatrix
            csr_factory = lambda rp, c, v, shape: CsrMatrix(rp, c, v, shape, self.device)
        self.spmm = spmm_fn
        rowptr, col, val = adjacency if adjacency is not None else normalized_adjacency_host(train_array, n_users, n_items)
        (urp, ucol, uval), (irp, icol, ival) = local_blocks_host(rowptr, col, val, self.layout, rank)
        L = self.layout
        self.local_nnz = int(urp[-1] + irp[-1])
        self.global_nnz = int(rowptr[-1])
        self.csr_u = csr_factory(urp, ucol, uval, (L.bu, L.n_pad))
        self.csr_i = csr_factory(irp, icol, ival, (L.bi, L.n_pad))
        self._bufs = None

    # ---- buffers -----------------------------------------------------------------
    def _buffers(self, d):
        if self._bufs is None or self._bufs[0].shape[1] != d:
            L = self.layout
            self._bufs = [torch.zeros((L.n_pad, d), dtype=torch.float32, device=self.device)
                          for _ in range(max(self.n_layers, 1))]
            self._rep_u = torch.empty((L.bu, d), dtype=torch.float32, device=self.device)
            self._rep_i = torch.empty((L.bi, d), dtype=torch.float32, device=self.device)
        return self._bufs

    def _slot(self, buf, part):
        L = self.layout
        if part == 'u':
            return buf[self.rank * L.bu:(self.rank + 1) * L.bu]
        return buf[L.pu + self.rank * L.bi: L.pu + (self.rank + 1) * L.bi]

    def _section(self, buf, part):
        L = self.layout
        return buf[:L.pu] if part == 'u' else buf[L.pu:]

    def _allgather(self, buf, part):
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            return None
        return dist.all_gather_into_tensor(self._section(buf, part), self._slot(buf, part), group=self.group,
                                           async_op=True)

    def load_local_embedding(self, emb_u_local, emb_i_local):
        """Owned rows of the layer-0 embeddings -> slot of X_0, then exchange X_0."""
        d = emb_u_local.shape[1]
        x0 = self._buffers(d)[0]
        su, si = self._slot(x0, 'u'), self._slot(x0, 'i')
        su.zero_(); si.zero_()
        su[:emb_u_local.shape[0]].copy_(emb_u_local)
        si[:emb_i_local.shape[0]].copy_(emb_i_local)
        for w in (self._allgather(x0, 'u'), self._allgather(x0, 'i')):
            if w is not None:
                w.wait()
        return x0

    # ---- the sharded K-layer pass -------------------------------------------------
    def propagate(self, x0=None):
        """mean(X_0..X_K) on the owned rows: (rep_users [bu, d], rep_items [bi, d]).
        `x0`: replicated padded layer-0 buffer (default: the one filled by
        load_local_embedding)."""
        bufs = self._buffers(x0.shape[1] if x0 is not None else self._bufs[0].shape[1])
        if x0 is not None and x0.data_ptr() != bufs[0].data_ptr():
            bufs[0].copy_(x0)
        K = self.n_layers
        s = 1.0 / (K + 1)
        if K == 0:
            return self._slot(bufs[0], 'u').clone(), self._slot(bufs[0], 'i').clone()
        pending = {}                                     # (layer, part) -> in-flight all-gather
        csr = {'u': self.csr_u, 'i': self.csr_i}
        rep = {'u': self._rep_u, 'i': self._rep_i}
        for l in range(K):
            last = l == K - 1
            src = bufs[l]
            order = ('u', 'i') if l % 2 == 0 else ('i', 'u')
            for part in order:
                other = 'i' if part == 'u' else 'u'
                w = pending.pop((l, other), None)        # this half reads X_l[other], replicated
                if w is not None:
                    w.wait()
                if last:
                    adds = [self._slot(bufs[j], part) for j in range(K)]
                    self.spmm(csr[part], src, out=rep[part], adds=adds, out_scale=s, add_scale=s)
                else:
                    dst = bufs[l + 1]
                    self.spmm(csr[part], src, out=self._slot(dst, part))
                    pending[(l + 1, part)] = self._allgather(dst, part)
        for w in pending.values():                        # nothing should be left; be safe
            if w is not None:
                w.wait()
        return rep['u'], rep['i']

    def gather_full_rep(self, rep_u, rep_i):
        """Replicated [n_users + n_items, d] from the owned blocks (used once per
        evaluation: scoring shards by user and needs every item row)."""
        L = self.layout
        d = rep_u.shape[1]
        full_u = torch.empty((L.pu, d), dtype=torch.float32, device=self.device)
        full_i = torch.empty((L.pi, d), dtype=torch.float32, device=self.device)
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            full_u.copy_(rep_u); full_i.copy_(rep_i)
        else:
            dist.all_gather_into_tensor(full_u, rep_u.contiguous(), group=self.group)
            dist.all_gather_into_tensor(full_i, rep_i.contiguous(), group=self.group)
        return torch.cat([full_u[:L.n_users], full_i[:L.n_items]], dim=0)


# ---------------------------------------------------------------------------------------
# Row-sharded LightGCN training: parameters, gradients and Adam state live on the owner rank.
# ---------------------------------------------------------------------------------------
class _ShardedPropagateFn(torch.autograd.Function):
    """Owned rows of E  ->  owned rows of mean(X_0..X_K).  The operator is
    P = mean_l A_hat^l with A_hat symmetric, so the backward pass is the same
    sharded pass applied to the incoming gradient (P^T = P)."""

    @staticmethod
    def forward(ctx, e_u, e_i, prop):
        ctx.prop = prop
        prop.load_local_embedding(e_u.detach(), e_i.detach())
        ru, ri = prop.propagate()
        return ru.clone(), ri.clone()

    @staticmethod
    def backward(ctx, g_u, g_i):
        prop = ctx.prop
        prop.load_local_embedding(g_u.contiguous(), g_i.contiguous())
        gu, gi = prop.propagate()
        return gu.clone(), gi.clone(), None


class _GatherRowsFn(torch.autograd.Function):
    """Owned blocks -> replicated [n_users + n_items, d].  Every rank evaluates the
    SAME full batch on the replicated rows, so each rank's gradient w.r.t. the
    replicated tensor is already the total: the backward pass just keeps the rows
    this rank owns (no reduce-scatter needed)."""

    @staticmethod
    def forward(ctx, blk_u, blk_i, prop):
        ctx.prop = prop
        ctx.nu, ctx.ni = blk_u.shape[0], blk_i.shape[0]
        return prop.gather_full_rep(blk_u.detach(), blk_i.detach())

    @staticmethod
    def backward(ctx, g_full):
        L, r = ctx.prop.layout, ctx.prop.rank
        (ulo, uhi), (ilo, ihi) = L.user_rows(r), L.item_rows(r)
        gu = g_full.new_zeros((ctx.nu, g_full.shape[1]))
        gi = g_full.new_zeros((ctx.ni, g_full.shape[1]))
        if uhi > ulo:
            gu[:uhi - ulo] = g_full[ulo:uhi]
        if ihi > ilo:
            gi[:ihi - ilo] = g_full[L.n_users + ilo: L.n_users + ihi]
        return gu, gi, None


class ShardedLightGCN(torch.nn.Module):
    """LightGCN (model.py:75-123) with the embedding table, its gradient and the
    optimizer state row-sharded over the ranks.  One step:
      X_0 exchange + K sharded half-layer passes  ->  owned rows of rep
      one all-gather of rep (and of E for the L2 term, model.py:110-113)
      the full batch's fused BPR loss on every rank (identical batch: same sampler seed)
      backward: owned rows of d loss / d rep  ->  the same sharded pass  ->  d loss / d E (owned rows)
      Adam on the owned rows.
    `loss_fn(rep_full, emb_full, users, pos, neg, n_users) -> tensor[2]` defaults to the
    fused HIP kernel (ops.bpr_loss_terms); the CPU tests inject a torch one."""

    def __init__(self, dataset, embedding_size, n_layers, rank, world, device, group=None, seed=2021,
                 spmm_fn=None, csr_factory=None, loss_fn=None, full_embedding=None):
        super().__init__()
        self.n_users, self.n_items, self.n_layers = dataset.n_users, dataset.n_items, n_layers
        self.prop = RowShardedPropagator(dataset.train_array, dataset.n_users, dataset.n_items, n_layers, rank, world,
                                         device, group=group, spmm_fn=spmm_fn, csr_factory=csr_factory)
        L = self.prop.layout
        (ulo, uhi), (ilo, ihi) = L.user_rows(rank), L.item_rows(rank)
        if full_embedding is None:                       # normal_(std=0.1), model.py:82 — same table on every rank
            g = torch.Generator(device='cpu').manual_seed(seed)
            full_embedding = torch.randn(self.n_users + self.n_items, embedding_size, generator=g) * 0.1
        eu = torch.zeros(L.bu, embedding_size)
        ei = torch.zeros(L.bi, embedding_size)
        eu[:max(uhi - ulo, 0)] = full_embedding[ulo:uhi]
        ei[:max(ihi - ilo, 0)] = full_embedding[self.n_users + ilo: self.n_users + ihi]
        self.emb_users = torch.nn.Parameter(eu.to(device))
        self.emb_items = torch.nn.Parameter(ei.to(device))
        if loss_fn is None:
            from . import ops
            loss_fn = lambda rep, emb, u, p, n, nu: ops.bpr_loss_terms(rep, rep, emb, emb, None, u, p, n, nu, nu)
        self.loss_fn = loss_fn

    def get_rep_local(self):
        return _ShardedPropagateFn.apply(self.emb_users, self.emb_items, self.prop)

    def bpr_loss_terms(self, users, pos_items, neg_items):
        ru, ri = self.get_rep_local()
        rep_full = _GatherRowsFn.apply(ru, ri, self.prop)
        emb_full = _GatherRowsFn.apply(self.emb_users, self.emb_items, self.prop)
        return self.loss_fn(rep_full, emb_full, users, pos_items, neg_items, self.n_users)

    def full_embedding(self):
        with torch.no_grad():
            return self.prop.gather_full_rep(self.emb_users.detach(), self.emb_items.detach())


# ---------------------------------------------------------------------------------------
# Embedding-column sharding: every rank holds the whole graph and d / P columns of every row.
# ---------------------------------------------------------------------------------------
class ColumnShardedLightGCN(torch.nn.Module):
    """LightGCN with the embedding COLUMNS cut over the ranks.

    Y[:, s] = A_hat X[:, s]: the columns of the embedding matrix propagate independently, so
    with the (small) CSR replicated on every GPU the K-layer pass needs NO exchange at all;
    each rank runs the ordinary single-GPU kernels on an [N, d/P] slice.  What does couple the
    slices is the scoring: a BPR dot product is the sum of the slices' partial dots (one
    all-reduce of 3*B floats per step), and evaluation needs whole rows (one all-gather of the
    final representation per evaluation).  Gradients, Adam state and parameters stay local.

    Why this and not row sharding at Amazon-book scale: a row-sharded layer must move
    (P-1)/P * N * d * 4 bytes into every GPU over xGMI (~0.3-0.4 TB/s aggregate) while the
    local SpMM reads nnz/P * (8 + 4d) bytes from HBM / Infinity Cache at 6-9 TB/s; with
    nnz/N ~ 21 the exchange is ~10x the compute.  Column slices of >= 16 floats keep every
    gather a whole 64-byte sector; at d/P = 8 half of each sector is wasted, which is the
    efficiency this mode gives up at P = 8, d = 64.  Row sharding (RowShardedPropagator) stays
    the choice when the graph itself outgrows one GPU's HBM.
    """

    def __init__(self, dataset, embedding_size, n_layers, rank, world, device, group=None, seed=2021,
                 full_embedding=None, adjacency=None, propagate_fn=None, loss_fn=None):
        super().__init__()
        if embedding_size % world:
            raise ValueError('embedding_size must be divisible by the number of ranks')
        self.n_users, self.n_items, self.n_layers = dataset.n_users, dataset.n_items, n_layers
        self.rank, self.world, self.group = rank, world, group
        self.d, self.dl = embedding_size, embedding_size // world
        n = self.n_users + self.n_items
        if full_embedding is None:
            g = torch.Generator(device='cpu').manual_seed(seed)
            full_embedding = torch.randn(n, embedding_size, generator=g) * 0.1
        self.emb = torch.nn.Parameter(full_embedding[:, rank * self.dl:(rank + 1) * self.dl].contiguous().to(device))
        if propagate_fn is None:
            from . import ops
            from .graph import CsrMatrix
            rowptr, col, val = adjacency if adjacency is not None else \
                normalized_adjacency_host(dataset.train_array, self.n_users, self.n_items)
            self.norm_adj = CsrMatrix(rowptr, col, val, (n, n), device)
            propagate_fn = lambda e: ops.PropagateFn.apply(e, self.norm_adj, self.norm_adj, n_layers)
        if loss_fn is None:
            from . import ops
            loss_fn = lambda rep, emb, u, p, i, nu, red: ops.bpr_loss_terms(rep, rep, emb, emb, None, u, p, i, nu, nu, red)
        self.propagate_fn, self.loss_fn = propagate_fn, loss_fn

    def _allreduce(self, t):
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(t, group=self.group)
        return t

    def get_rep_local(self):
        return self.propagate_fn(self.emb)

    def bpr_loss_terms(self, users, pos_items, neg_items):
        return self.loss_fn(self.get_rep_local(), self.emb, users, pos_items, neg_items, self.n_users, self._allreduce)

    def gather_columns(self, local):
        """[N, d/P] slices -> replicated [N, d] (once per evaluation)."""
        if not (dist.is_available() and dist.is_initialized()):
            return local.clone()
        n = local.shape[0]
        parts = torch.empty((self.world * n, self.dl), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(parts, local.contiguous(), group=self.group)
        return parts.view(self.world, n, self.dl).permute(1, 0, 2).reshape(n, self.d).contiguous()

    def full_embedding(self):
        with torch.no_grad():
            return self.gather_columns(self.emb.detach())


def column_shard_model(model_config, dataset, rank, world, group=None, full_state=None, reduce_fn=None):
    """Any of MF / LightGCN / IGCN / IMF with its embedding COLUMNS cut over the ranks: the model is
    built with embedding_size / world columns (graph, template features and every kernel unchanged),
    takes its column slice of `full_state` (a full-width state dict; default: a seeded full-width
    initialisation, identical on every rank) and sums the partial dots of its losses over the ranks.
    Propagation needs no exchange (see ColumnShardedLightGCN)."""
    from .model import get_model
    d = model_config['embedding_size']
    if d % world:
        raise ValueError('embedding_size must be divisible by the number of ranks')
    dl = d // world
    if full_state is None:
        g_state = torch.random.get_rng_state()
        torch.manual_seed(model_config.get('seed', 2021))
        full = get_model(model_config, dataset)
        full_state = {k: v.detach().clone() for k, v in full.state_dict().items()}
        del full
        torch.random.set_rng_state(g_state)
    model = get_model(dict(model_config, embedding_size=dl), dataset)
    sl = slice(rank * dl, (rank + 1) * dl)
    with torch.no_grad():
        for name, p in model.state_dict().items():
            src = full_state[name]
            p.copy_(src[..., sl] if src.shape[-1] == d else src)
    if reduce_fn is None:
        def reduce_fn(t):
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(t, group=group)
            return t
    model.slice_reduce_fn = reduce_fn
    return model

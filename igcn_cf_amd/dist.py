"""Row-sharded propagation and training across the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Not in the reference (single process, single device — SURVEY.md section 2.1); this is
the scheme BASELINE.json's north_star asks for at Amazon-book scale and above:

* the user set and the item set are each cut into P contiguous blocks balanced by NONZEROS
  (ShardLayout.balanced: boundaries from the prefix sums of the row pointers; power-law item
  degrees make equal row counts unequal work); rank r owns user block r and item block r:
  those rows of A_hat as local CSR (padded global column ids), of the embeddings, of every
  layer's output, of the gradients and of the optimizer state;
* every layer input X_l is replicated (the all-gather target).  Two replicated buffers are
  used in turn; the rows a rank computes go to its own [rows, d] buffer of that layer (the
  layer mean needs them again) and from there into the other ranks' replicated buffers;
* exchange, chosen by size:
  - 'fused' (the replicated operand is small, latency decides — Amazon-book: 52.8 MB): padded
    layout [rank 0: users, items | rank 1: users, items | ...], ONE local SpMM over the
    rank's user and item rows and ONE all-gather per layer;
  - 'halves' (the operand is GBs, bandwidth decides — BASELINE config 5): padded layout
    [all user blocks | all item blocks]; the graph is bipartite (user rows read only item
    embeddings and vice versa), so a layer is two half-steps whose order alternates from
    layer to layer and every all-gather runs under the SpMM of the other half;
* the layer mean is the epilogue of the last local SpMM, on owned rows only;
* a training step needs 3 B rows of the result (and of the raw embeddings): they are exchanged
  as ONE all-reduce of a [3 B, 2 d] buffer in which every rank fills the rows it owns
  (x + 0 is exact), never the full tables.

`spmm_fn` is the local product; the product path uses the HIP kernel (ops.spmm).  The CPU
tests inject a checker implementation to exercise the partitioning / exchange logic under
gloo with world_size 2.  Nothing here has run on more than one physical GPU yet (gpurun boxes have one).
"""
import numpy as np
import torch
import torch.distributed as dist

from .graph import normalized_adjacency_host

FUSED_EXCHANGE_MAX_BYTES = 128 << 20       # replicated operand (at d = 64) up to which one all-gather per layer is used


def _dist_on():
    return dist.is_available() and dist.is_initialized()


class TorchCollectives:
    """The collectives of the sharded path: torch.distributed over the process group (RCCL on the GPUs, gloo in the CPU
    tests).  A seam, not an abstraction layer: tests pass an object of the same shape whose copies complete as LATE as RCCL's
    may (at wait(), tests/test_dist_cpu.py) — gloo's wait() blocks the host, so a missing wait goes unnoticed under it."""

    def __init__(self, group=None):
        self.group = group

    def active(self):
        return _dist_on()

    def all_gather_into_tensor(self, out, inp, async_op):
        return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)

    def all_reduce(self, buf):
        dist.all_reduce(buf, group=self.group)


class ShardLayout:
    """Contiguous user / item blocks per rank and the padded layout of the replicated buffers.

    user_bounds / item_bounds: int64 [world + 1] block boundaries (default: equal row counts).
    bu / bi: padded block sizes (the largest block).  fused: see the module docstring."""

    def __init__(self, n_users, n_items, world, user_bounds=None, item_bounds=None, fused=False):
        self.n_users, self.n_items, self.world, self.fused = int(n_users), int(n_items), int(world), bool(fused)
        eq = lambda n: np.minimum(np.arange(world + 1, dtype=np.int64) * ((n + world - 1) // world), n)
        self.user_bounds = eq(self.n_users) if user_bounds is None else np.asarray(user_bounds, dtype=np.int64)
        self.item_bounds = eq(self.n_items) if item_bounds is None else np.asarray(item_bounds, dtype=np.int64)
        for bnd, n in ((self.user_bounds, self.n_users), (self.item_bounds, self.n_items)):
            if bnd.shape != (world + 1,) or bnd[0] != 0 or bnd[-1] != n or np.any(np.diff(bnd) < 0):
                raise ValueError('block boundaries must rise from 0 to the section size')
        self.bu = max(1, int(np.diff(self.user_bounds).max()))
        self.bi = max(1, int(np.diff(self.item_bounds).max()))
        self.pu, self.pi = self.bu * world, self.bi * world          # padded section sizes
        self.n_pad = self.pu + self.pi
        self.block = self.bu + self.bi                                # rows a rank owns (padded)

    @classmethod
    def balanced(cls, rowptr, n_users, n_items, world, fused=False):
        """Blocks with (nearly) equal nonzeros, separately within the user and the item range."""
        rowptr = np.asarray(rowptr, dtype=np.int64)

        def cut(lo, hi):
            nnz = rowptr[lo:hi + 1] - rowptr[lo]
            targets = nnz[-1] * np.arange(1, world, dtype=np.float64) / world
            inner = np.searchsorted(nnz, targets, side='left')
            return np.concatenate([[0], np.clip(inner, 0, hi - lo), [hi - lo]]).astype(np.int64)
        ub, ib = cut(0, n_users), cut(n_users, n_users + n_items)
        return cls(n_users, n_items, world, np.maximum.accumulate(ub), np.maximum.accumulate(ib), fused)

    def user_rows(self, rank):
        return int(self.user_bounds[rank]), int(self.user_bounds[rank + 1])

    def item_rows(self, rank):
        return int(self.item_bounds[rank]), int(self.item_bounds[rank + 1])

    def owner(self, node):
        """(rank, row inside the rank's padded block [bu users ; bi items]) of global node ids."""
        node = np.asarray(node, dtype=np.int64)
        is_item = node >= self.n_users
        it = np.where(is_item, node - self.n_users, 0)
        us = np.where(is_item, 0, node)
        ru = np.searchsorted(self.user_bounds, us, side='right') - 1
        ri = np.searchsorted(self.item_bounds, it, side='right') - 1
        ru, ri = np.minimum(ru, self.world - 1), np.minimum(ri, self.world - 1)
        rank = np.where(is_item, ri, ru)
        local = np.where(is_item, self.bu + it - self.item_bounds[ri], us - self.user_bounds[ru])
        return rank, local

    def pad_index(self, node):
        """global node id (users first, then items) -> row in the padded replicated layout."""
        rank, local = self.owner(node)
        if self.fused:
            return rank * self.block + local
        is_item = np.asarray(node, dtype=np.int64) >= self.n_users
        return np.where(is_item, self.pu + rank * self.bi + (local - self.bu), rank * self.bu + local)


    def pad_index_torch(self, node):
        """pad_index for an int64 tensor of global node ids, on the tensor's device (the device builders of the local blocks)."""
        dev = node.device
        ub = torch.from_numpy(self.user_bounds).to(dev)
        ib = torch.from_numpy(self.item_bounds).to(dev)
        is_item = node >= self.n_users
        it = torch.where(is_item, node - self.n_users, torch.zeros_like(node))
        us = torch.where(is_item, torch.zeros_like(node), node)
        ru = (torch.searchsorted(ub, us, right=True) - 1).clamp_(max=self.world - 1)
        ri = (torch.searchsorted(ib, it, right=True) - 1).clamp_(max=self.world - 1)
        lu, li = us - ub[ru], it - ib[ri]
        if self.fused:
            return torch.where(is_item, ri * self.block + self.bu + li, ru * self.block + lu)
        return torch.where(is_item, self.pu + ri * self.bi + li, ru * self.bu + lu)


def local_blocks_host(rowptr, col, val, layout, rank):
    """Rows of the global CSR owned by `rank`, as two CSR blocks (user rows, item rows) of
    exactly bu / bi rows (padding rows empty), padded column ids."""
    out = []
    for (lo, hi), nb, base in ((layout.user_rows(rank), layout.bu, 0),
                               (layout.item_rows(rank), layout.bi, layout.n_users)):
        s, e = rowptr[base + lo], rowptr[base + hi]
        rp = np.full(nb + 1, e - s, dtype=np.int64)
        rp[:hi - lo + 1] = rowptr[base + lo: base + hi + 1] - s
        out.append((rp, layout.pad_index(col[s:e]).astype(np.int32), val[s:e].copy()))
    return out


class RowShardedPropagator:
    """adjacency / train_array: the whole A_hat on the host, cut here (Amazon-book size: 36 MB).  At sizes where no rank can
    hold the whole matrix (BASELINE config 5: 1 G nonzeros) pass `layout` + `local_blocks` instead — the rank's own blocks,
    already in HBM with padded column ids (synth.BipartiteGraphDevice.rank_blocks): (csr_users, csr_items) for 'halves',
    (csr,) for 'fused' — and `global_nnz`; nothing of the whole graph is then touched here."""

    def __init__(self, train_array, n_users, n_items, n_layers, rank, world, device, group=None,
                 spmm_fn=None, csr_factory=None, adjacency=None, exchange=None, balance=True,
                 layout=None, local_blocks=None, global_nnz=None, collectives=None):
        self.collectives = collectives if collectives is not None else TorchCollectives(group)
        if exchange is None:
            exchange = 'fused' if (n_users + n_items) * 256 <= FUSED_EXCHANGE_MAX_BYTES else 'halves'
        if exchange not in ('fused', 'halves'):
            raise ValueError("exchange must be 'fused' or 'halves'")
        self.exchange = exchange
        fused = exchange == 'fused'
        self.n_layers, self.rank, self.world, self.group = n_layers, rank, world, group
        self.device = torch.device(device)
        if spmm_fn is None:
            from . import ops
            spmm_fn = ops.spmm
        self.spmm = spmm_fn
        self._d = None
        if local_blocks is not None:
            if layout is None or layout.fused != fused or layout.world != world:
                raise ValueError('local_blocks need the ShardLayout they were cut with (same world, same exchange)')
            if len(local_blocks) != (1 if fused else 2):
                raise ValueError("local_blocks: (csr,) for 'fused', (csr_users, csr_items) for 'halves'")
            self.layout = L = layout
            shapes = [(L.block, L.n_pad)] if fused else [(L.bu, L.n_pad), (L.bi, L.n_pad)]
            for blk, shape in zip(local_blocks, shapes):
                if tuple(blk.shape) != shape:
                    raise ValueError('local block of shape %s, layout needs %s' % (tuple(blk.shape), shape))
            if fused:
                self.csr, = local_blocks
            else:
                self.csr_u, self.csr_i = local_blocks
            self.local_nnz = int(sum(blk.nnz for blk in local_blocks))
            self.global_nnz = None if global_nnz is None else int(global_nnz)
            return
        rowptr, col, val = adjacency if adjacency is not None else normalized_adjacency_host(train_array, n_users, n_items)
        self.layout = ShardLayout.balanced(rowptr, n_users, n_items, world, fused) if balance else \
            ShardLayout(n_users, n_items, world, fused=fused)
        if csr_factory is None:
            from .graph import XCD_PLAN, CsrMatrix
            # 'fused' (operand of tens of MB, slices of it fit an XCD's L2): the XCD plan, as on one GPU; 'halves'
            # (operand of GBs): the plain long-row plan — cutting rows buys nothing when a slice is 100x an L2
            plan = XCD_PLAN if exchange == 'fused' else None
            csr_factory = lambda rp, c, v, shape, blocks=None: CsrMatrix(rp, c, v, shape, self.device, order_blocks=blocks,
                                                                         xcd_plan=plan)
        (urp, ucol, uval), (irp, icol, ival) = local_blocks_host(rowptr, col, val, self.layout, rank)
        L = self.layout
        self.local_nnz = int(urp[-1] + irp[-1])
        self.global_nnz = int(rowptr[-1])
        if fused:                                        # one matrix: the rank's user rows, then its item rows
            rp = np.concatenate([urp, irp[1:] + urp[-1]])
            self.csr = self._make_csr(csr_factory, rp, np.concatenate([ucol, icol]), np.concatenate([uval, ival]),
                                      (L.block, L.n_pad), [0, L.bu, L.block])
        else:
            self.csr_u = self._make_csr(csr_factory, urp, ucol, uval, (L.bu, L.n_pad), [0, L.bu])
            self.csr_i = self._make_csr(csr_factory, irp, icol, ival, (L.bi, L.n_pad), [0, L.bi])

    @staticmethod
    def _make_csr(factory, rp, col, val, shape, blocks):
        try:
            return factory(rp, col, val, shape, blocks)          # rows dealt to the waves by length inside each phase
        except TypeError:
            return factory(rp, col, val, shape)                  # injected 4-argument factories (CPU tests)

    # ---- buffers -----------------------------------------------------------------
    def _buffers(self, d):
        """own[l]: this rank's rows of table l of ops.mean_plan ([block, d]; user rows first; table 0 = X_0, table K = the
        mean), l = 0..K; rep[0..1]: the two
        replicated buffers used in turn."""
        if self._d != d:
            L, K = self.layout, self.n_layers
            z = lambda rows: torch.zeros((rows, d), dtype=torch.float32, device=self.device)
            self._own = [z(L.block) for _ in range(K + 1)]
            self._rep = [z(L.n_pad) for _ in range(2 if K > 1 else 1)]
            self._d = d
        return self._own, self._rep

    def _part(self, own, part):
        L = self.layout
        return own[:L.bu] if part == 'u' else own[L.bu:]

    def _section(self, rep, part):
        L = self.layout
        return rep[:L.pu] if part == 'u' else rep[L.pu:]

    def _allgather(self, out, inp, async_op):
        if not self.collectives.active():
            out.copy_(inp)                               # one rank without a process group
            return None
        return self.collectives.all_gather_into_tensor(out, inp, async_op)

    def load_local_embedding(self, emb_u_local, emb_i_local):
        """Owned rows of the layer-0 embeddings -> own[0], then exchange X_0 into rep[0]."""
        d = emb_u_local.shape[1]
        own, rep = self._buffers(d)
        L = self.layout
        own[0].zero_()
        own[0][:emb_u_local.shape[0]].copy_(emb_u_local)
        own[0][L.bu:L.bu + emb_i_local.shape[0]].copy_(emb_i_local)
        if self.exchange == 'fused':
            self._allgather(rep[0], own[0], False)
        else:
            ws = [self._allgather(self._section(rep[0], p), self._part(own[0], p), True) for p in ('u', 'i')]
            for w in ws:
                if w is not None:
                    w.wait()
        return rep[0]

    # ---- the sharded K-layer pass -------------------------------------------------
    def propagate(self, plan=None):
        """mean(X_0..X_K) on the owned rows: (rep_users [bu, d], rep_items [bi, d]) — views of one
        [block, d] buffer that the next call overwrites.  X_0 is what load_local_embedding left.  The K launches are
        those of ops.mean_plan (own[l] = the owned rows of its table l), as on one GPU: every launch's output but the
        last is exchanged, so the number and size of the all-gathers is that of the plain layer loop."""
        own, rep = self._buffers(self._d)
        K, L = self.n_layers, self.layout
        s = 1.0 / (K + 1)
        if K == 0:
            return self._part(own[0], 'u'), self._part(own[0], 'i')
        from .ops import mean_plan, plan_addends
        if plan is None:
            plan = mean_plan(K)
        if self.exchange == 'fused':
            for l, add in enumerate(plan):               # own[l + 1] = A own-rows-of(table l) (+ own[add]), ops.mean_plan
                src = rep[l % 2]
                adds = plan_addends(add, own)
                if l == K - 1:
                    self.spmm(self.csr, src, out=own[K], adds=adds, out_scale=s, add_scale=s)
                else:
                    self.spmm(self.csr, src, out=own[l + 1], adds=adds)
                    self._allgather(rep[(l + 1) % 2], own[l + 1], False)
            return self._part(own[K], 'u'), self._part(own[K], 'i')
        pending = {}                                     # (layer, part) -> in-flight all-gather
        csr = {'u': self.csr_u, 'i': self.csr_i}
        for l in range(K):
            last = l == K - 1
            src = rep[l % 2]
            order = ('u', 'i') if l % 2 == 0 else ('i', 'u')
            for part in order:
                other = 'i' if part == 'u' else 'u'
                w = pending.pop((l, other), None)        # this half reads X_l[other], replicated
                if w is not None:
                    w.wait()
                adds = [self._part(o, part) for o in plan_addends(plan[l], own)]
                if last:
                    self.spmm(csr[part], src, out=self._part(own[K], part), adds=adds, out_scale=s, add_scale=s)
                else:
                    # the half of table l+1 computed here is read by the OTHER half of launch l+1; the replicated buffer
                    # it goes to was last read by launch l-1, whose two halves are done
                    self.spmm(csr[part], src, out=self._part(own[l + 1], part), adds=adds)
                    pending[(l + 1, part)] = self._allgather(self._section(rep[(l + 1) % 2], part),
                                                             self._part(own[l + 1], part), True)
        for w in pending.values():                        # nothing should be left; be safe
            if w is not None:
                w.wait()
        return self._part(own[K], 'u'), self._part(own[K], 'i')

    def gather_full_rep(self, rep_u, rep_i):
        """Replicated [n_users + n_items, d] in the ORIGINAL node order from the owned blocks (evaluation:
        scoring shards by user and needs every item row)."""
        L = self.layout
        d = rep_u.shape[1]
        full_u = torch.empty((L.pu, d), dtype=torch.float32, device=self.device)
        full_i = torch.empty((L.pi, d), dtype=torch.float32, device=self.device)
        self._allgather(full_u, rep_u.contiguous(), False)
        self._allgather(full_i, rep_i.contiguous(), False)
        parts = [full_u[r * L.bu: r * L.bu + (L.user_bounds[r + 1] - L.user_bounds[r])] for r in range(L.world)]
        parts += [full_i[r * L.bi: r * L.bi + (L.item_bounds[r + 1] - L.item_bounds[r])] for r in range(L.world)]
        return torch.cat(parts, dim=0)


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


class _BatchRowsFn(torch.autograd.Function):
    """Rows `ids` (global node ids, any owner) of several row-sharded tables -> replicated [M, d] each.

    Every rank writes the rows it owns into a zero [M, sum d] buffer and ONE all-reduce (sum) completes it
    — M = 3 B rows per training step instead of the whole table.  Every rank evaluates the same batch on
    the replicated rows, so each rank's gradient w.r.t. them is already the total: the backward pass adds
    the rows a rank owns into its local gradient tables (duplicate ids add up), no communication."""

    @staticmethod
    def forward(ctx, prop, ids, *tables_ui):
        L, rank = prop.layout, prop.rank
        n_tab = len(tables_ui) // 2
        widths = [tables_ui[2 * t].shape[1] for t in range(n_tab)]
        ctx.widths, ctx.shapes = widths, [t.shape for t in tables_ui]
        ctx.device_path = ids.is_cuda
        if ids.is_cuda:
            # two launches per table pair, no host round trip: the rank's own rows, zeros elsewhere
            from . import _lib
            (ulo, uhi), (ilo, ihi) = L.user_rows(rank), L.item_rows(rank)
            ctx.ids, ctx.bounds = ids, (L.n_users, ulo, uhi, ilo, ihi)
            buf = torch.empty((ids.numel(), sum(widths)), dtype=torch.float32, device=ids.device)
            c0 = 0
            for t in range(n_tab):
                tu, ti = tables_ui[2 * t].detach(), tables_ui[2 * t + 1].detach()
                _lib.check(_lib.lib().igcn_owned_rows_gather_f32(
                    ids.data_ptr(), ids.numel(), L.n_users, ulo, uhi, ilo, ihi, tu.data_ptr(), tu.stride(0), ti.data_ptr(),
                    ti.stride(0), widths[t], buf.data_ptr() + 4 * c0, buf.stride(0), _lib.current_stream()),
                    'igcn_owned_rows_gather_f32')
                c0 += widths[t]
        else:                                            # host tensors: the CPU tests of the exchange logic
            o, l = L.owner(ids.numpy())
            owner, local = torch.from_numpy(o), torch.from_numpy(l)
            sel = torch.nonzero(owner == rank, as_tuple=False).flatten()
            loc = local[sel]
            buf = torch.zeros((ids.numel(), sum(widths)), dtype=torch.float32, device=tables_ui[0].device)
            c0 = 0
            for t in range(n_tab):
                tu, ti = tables_ui[2 * t].detach(), tables_ui[2 * t + 1].detach()
                is_item = loc >= L.bu
                rows = torch.where(is_item[:, None], ti[(loc - L.bu).clamp(min=0, max=max(ti.shape[0] - 1, 0))],
                                   tu[loc.clamp(max=max(tu.shape[0] - 1, 0))])
                buf[sel, c0:c0 + widths[t]] = rows
                c0 += widths[t]
            ctx.sel, ctx.loc, ctx.bu = sel, loc, L.bu
        if prop.collectives.active():
            prop.collectives.all_reduce(buf)
        outs, c0 = [], 0
        for w in widths:
            outs.append(buf[:, c0:c0 + w])
            c0 += w
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        out = []
        if ctx.device_path:
            from . import _lib
            n_users, ulo, uhi, ilo, ihi = ctx.bounds
            for t, g in enumerate(grads):
                gu = torch.zeros(ctx.shapes[2 * t], dtype=torch.float32, device=ctx.ids.device)
                gi = torch.zeros(ctx.shapes[2 * t + 1], dtype=torch.float32, device=ctx.ids.device)
                if g is not None:
                    if g.stride(1) != 1:
                        g = g.contiguous()
                    _lib.check(_lib.lib().igcn_owned_rows_scatter_add_f32(
                        ctx.ids.data_ptr(), ctx.ids.numel(), n_users, ulo, uhi, ilo, ihi, g.data_ptr(), g.stride(0),
                        ctx.widths[t], gu.data_ptr(), gu.stride(0), gi.data_ptr(), gi.stride(0), _lib.current_stream()),
                        'igcn_owned_rows_scatter_add_f32')
                out += [gu, gi]
            return (None, None, *out)
        sel, loc, bu = ctx.sel, ctx.loc, ctx.bu
        is_item = loc >= bu
        su, si = sel[~is_item], sel[is_item]
        lu, li = loc[~is_item], loc[is_item] - bu
        for t, g in enumerate(grads):
            gu = g.new_zeros(ctx.shapes[2 * t])
            gi = g.new_zeros(ctx.shapes[2 * t + 1])
            if g is not None:
                gu.index_add_(0, lu, g[su])
                gi.index_add_(0, li, g[si])
            out += [gu, gi]
        return (None, None, *out)


def _compact_bpr_terms(rows_rep, rows_emb, batch):
    """The fused BPR kernel on the exchanged rows: [users | positives | negatives], B rows each."""
    from . import ops
    idx = torch.arange(batch, dtype=torch.int64, device=rows_rep.device)
    rr, re = rows_rep.contiguous(), rows_emb.contiguous()
    return ops.bpr_loss_terms(rr, rr, re, re, None, idx, idx, idx + batch, batch, batch)


class ShardedLightGCN(torch.nn.Module):
    """LightGCN (model.py:75-123) with the embedding table, its gradient and the
    optimizer state row-sharded over the ranks.  One step:
      X_0 exchange + K sharded layers                ->  owned rows of rep
      ONE all-reduce of the 3 B batch rows of rep and of E (model.py:110-116 reads nothing else)
      the batch's fused BPR loss on every rank (identical batch: same sampler seed)
      backward: owned batch rows of d loss / d rep   ->  the same sharded pass  ->  d loss / d E (owned rows)
      Adam on the owned rows.
    `loss_fn(rows_rep [3B, d], rows_emb [3B, d], B) -> tensor[2]` (rows: users, positives, negatives) defaults
    to the fused HIP kernel; the CPU tests inject a torch one."""

    def __init__(self, dataset, embedding_size, n_layers, rank, world, device, group=None, seed=2021,
                 spmm_fn=None, csr_factory=None, loss_fn=None, full_embedding=None, exchange=None, balance=True,
                 adjacency=None, collectives=None):
        super().__init__()
        self.n_users, self.n_items, self.n_layers = dataset.n_users, dataset.n_items, n_layers
        self.prop = RowShardedPropagator(dataset.train_array, dataset.n_users, dataset.n_items, n_layers, rank, world,
                                         device, group=group, spmm_fn=spmm_fn, csr_factory=csr_factory, exchange=exchange,
                                         balance=balance, adjacency=adjacency, collectives=collectives)
        L = self.prop.layout
        (ulo, uhi), (ilo, ihi) = L.user_rows(rank), L.item_rows(rank)
        if full_embedding is None:                       # normal_(std=0.1), model.py:82 — same table on every rank
            g = torch.Generator(device='cpu').manual_seed(seed)
            full_embedding = torch.randn(self.n_users + self.n_items, embedding_size, generator=g) * 0.1
        eu = torch.zeros(L.bu, embedding_size)
        ei = torch.zeros(L.bi, embedding_size)
        eu[:uhi - ulo] = full_embedding[ulo:uhi]
        ei[:ihi - ilo] = full_embedding[self.n_users + ilo: self.n_users + ihi]
        self.emb_users = torch.nn.Parameter(eu.to(device))
        self.emb_items = torch.nn.Parameter(ei.to(device))
        self.loss_fn = loss_fn if loss_fn is not None else _compact_bpr_terms

    def get_rep_local(self):
        return _ShardedPropagateFn.apply(self.emb_users, self.emb_items, self.prop)

    def bpr_loss_terms(self, users, pos_items, neg_items):
        ru, ri = self.get_rep_local()
        ids = torch.cat([users, self.n_users + pos_items, self.n_users + neg_items])
        rows_rep, rows_emb = _BatchRowsFn.apply(self.prop, ids, ru, ri, self.emb_users, self.emb_items)
        return self.loss_fn(rows_rep, rows_emb, users.numel())

    def full_embedding(self):
        with torch.no_grad():
            return self.prop.gather_full_rep(self.emb_users.detach(), self.emb_items.detach())

    def recommend_local(self, k, excl=None):
        """Top-k item ids for the users THIS rank owns (global user ids [ulo, uhi)): the item rows of the
        representation are gathered once, every rank scores its own users against all of them (no collective
        in the scoring loop).  excl: host CSR (rowptr, col) of masked items over ALL users, or None."""
        from . import ops
        L, rank = self.prop.layout, self.prop.rank
        with torch.no_grad():
            ru, ri = self.get_rep_local()
            d = ri.shape[1]
            full_i = torch.empty((L.pi, d), dtype=torch.float32, device=ri.device)
            self.prop._allgather(full_i, ri.contiguous(), False)
            items = torch.cat([full_i[r * L.bi: r * L.bi + (L.item_bounds[r + 1] - L.item_bounds[r])] for r in range(L.world)])
            ulo, uhi = L.user_rows(rank)
            if uhi == ulo:
                return torch.empty((0, k), dtype=torch.int64, device=ri.device)
            rp = cl = None
            if excl is not None:
                rowptr, col = excl
                s, e = int(rowptr[ulo]), int(rowptr[uhi])
                rp = torch.from_numpy(np.ascontiguousarray(rowptr[ulo:uhi + 1] - s, dtype=np.int64)).to(ri.device)
                cl = torch.from_numpy(np.ascontiguousarray(col[s:e], dtype=np.int32)).to(ri.device)
            idx, _ = ops.score_topk(ru[:uhi - ulo].contiguous(), items.contiguous(), k, excl_rowptr=rp, excl_col=cl)
            return idx


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

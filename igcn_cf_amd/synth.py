"""Large synthetic bipartite interaction graphs generated in HBM (BASELINE config 5: 10 M users x 2 M items x 500 M
edges) and their per-rank shares under row sharding.

Same generator rules as dataset.SyntheticDataset: per-user interaction counts
~ max(min_inter, LogNormal) scaled to the requested total, items drawn from a Zipf-Mandelbrot popularity over a random
permutation of the item ids, de-duplicated per user — but with torch ops on the device, because half a billion pairs do
not go through numpy in a benchmark's time.  No split into train / val / test: these graphs feed the propagation only.

A_hat of such a graph (1 G nonzeros) is never built whole: a rank's share — its nnz-balanced block of user rows and
its block of item rows (dist.ShardLayout.balanced), global column ids, values d_r^-1/2 d_c^-1/2 (model.py:85-94;
duplicates are removed, so every stored entry of A is 1) — comes straight from the pair list and the two degree vectors.
"""
import numpy as np
import torch

from .graph import CsrMatrix


class BipartiteGraphDevice:
    def __init__(self, n_users, n_items, n_edges, device, seed=2021, min_inter=7, sigma=1.0, zipf_a=1.0, zipf_q=None):
        self.n_users, self.n_items, self.device = int(n_users), int(n_items), torch.device(device)
        if zipf_q is None:
            zipf_q = 150.0 * n_items / 96421.          # the Amazon-like preset's head share (~0.1 % of the edges per top item / log)
        g = torch.Generator(device=self.device).manual_seed(seed)
        z = torch.randn(self.n_users, device=self.device, generator=g) * sigma
        lo, hi = -5., 14.
        for _ in range(50):                             # scale of the log-normal by bisection on the total
            mu = 0.5 * (lo + hi)
            total = int(torch.clamp(torch.exp(mu + z).round(), min=min_inter).sum())
            lo, hi = (lo, mu) if total > n_edges else (mu, hi)
        cnt = torch.clamp(torch.exp(mu + z).round(), min=min_inter).to(torch.int64).clamp_(max=self.n_items // 2)
        rank = torch.arange(1, self.n_items + 1, device=self.device, dtype=torch.float64)
        cdf = torch.cumsum(1. / torch.pow(rank + zipf_q, zipf_a), 0)
        cdf /= cdf[-1].clone()
        perm = torch.randperm(self.n_items, device=self.device, generator=g)
        users = torch.repeat_interleave(torch.arange(self.n_users, device=self.device), cnt)
        draws = torch.rand(users.shape[0], device=self.device, generator=g, dtype=torch.float64)
        items = perm[torch.searchsorted(cdf, draws).clamp_(max=self.n_items - 1)]
        del draws, cdf, rank
        key = torch.unique(users * self.n_items + items)                 # per-user de-duplication; sorted by (user, item)
        del users, items
        self.users = torch.div(key, self.n_items, rounding_mode='floor')
        self.items = key - self.users * self.n_items
        del key
        self.n_edges = int(self.users.shape[0])
        self.deg_u = torch.bincount(self.users, minlength=self.n_users)
        self.deg_i = torch.bincount(self.items, minlength=self.n_items)
        self.n = self.n_users + self.n_items
        self.nnz = 2 * self.n_edges                                      # of A_hat

    def rowptr_host(self):
        """Row pointers of the whole A_hat (user rows, then item rows) on the host — what ShardLayout.balanced reads."""
        deg = torch.cat([self.deg_u, self.deg_i]).cpu().numpy()
        rowptr = np.zeros(self.n + 1, dtype=np.int64)
        np.cumsum(deg, out=rowptr[1:])
        return rowptr

    def inv_sqrt_degree(self):
        """D^-1/2 with deg = max(1, row sum) (model.py:87-89), float32 — from a table of numpy float32 powers, as
        graph.normalized_adjacency_device does, so that the values are bit-identical to the host builder's."""
        deg = torch.cat([self.deg_u, self.deg_i])
        max_deg = int(deg.max()) if deg.numel() else 1
        table = np.power(np.maximum(np.arange(max_deg + 1, dtype=np.float32), np.float32(1.)), np.float32(-0.5)).astype(np.float32)
        return torch.from_numpy(table).to(self.device)[deg]

    def _share_arrays(self, layout, rank):
        """The rank's two row blocks with GLOBAL column ids: ((counts, col int64, val) of its user rows — which gather item
        rows —, the same of its item rows).  Values fl(fl(d_r * 1) * d_c), the reference's order (model.py:90-92)."""
        (ulo, uhi), (ilo, ihi) = layout.user_rows(rank), layout.item_rows(rank)
        dinv = self.inv_sqrt_degree()
        n = self.n
        # user rows: the pair list is sorted by (user, item): the rank's pairs are one contiguous run
        bounds = torch.searchsorted(self.users, torch.tensor([ulo, uhi], device=self.device))
        s, e = int(bounds[0]), int(bounds[1])
        u_cols = self.items[s:e] + self.n_users
        u_val = (dinv[self.users[s:e]] * 1.0) * dinv[u_cols]
        # item rows: pairs whose item lies in the block, sorted by (item, user)
        m = (self.items >= ilo) & (self.items < ihi)
        key = (self.items[m] - ilo) * n + self.users[m]
        key = torch.sort(key).values
        i_rows = torch.div(key, n, rounding_mode='floor')
        i_cols = key - i_rows * n
        del key, m
        i_val = (dinv[self.n_users + ilo + i_rows] * 1.0) * dinv[i_cols]
        return (self.deg_u[ulo:uhi], u_cols, u_val), (self.deg_i[ilo:ihi], i_cols, i_val)

    def rank_share(self, layout, rank, xcd_plan=None):
        """(CsrMatrix of the rank's rows [its user block; its item block] with GLOBAL column ids, int64 global row ids)."""
        (ulo, uhi), (ilo, ihi) = layout.user_rows(rank), layout.item_rows(rank)
        (u_cnt, u_cols, u_val), (i_cnt, i_cols, i_val) = self._share_arrays(layout, rank)
        nu_l, ni_l = uhi - ulo, ihi - ilo
        rowptr = torch.zeros(nu_l + ni_l + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(torch.cat([u_cnt, i_cnt]), 0, out=rowptr[1:])
        col = torch.cat([u_cols, i_cols]).to(torch.int32)
        val = torch.cat([u_val, i_val])
        del u_cols, i_cols, u_val, i_val
        grow = torch.cat([torch.arange(ulo, uhi, device=self.device), self.n_users + torch.arange(ilo, ihi, device=self.device)])
        csr = CsrMatrix.from_device(rowptr, col, val, (nu_l + ni_l, self.n), order_blocks=[0, nu_l, nu_l + ni_l], xcd_plan=xcd_plan)
        return csr, grow

    def rank_blocks(self, layout, rank, csr_factory=None):
        """What dist.RowShardedPropagator(local_blocks=...) takes, built in HBM from the pair list — never a CSR of the
        whole graph, on the host or anywhere: the rank's user block [bu, n_pad] and item block [bi, n_pad] (rows padded to
        the layout's block sizes, column ids in the PADDED replicated layout), or, for a 'fused' layout, the one matrix
        [bu + bi, n_pad] of both.  csr_factory(rowptr, col int32, val, shape, order_blocks): default CsrMatrix.from_device."""
        (ulo, uhi), (ilo, ihi) = layout.user_rows(rank), layout.item_rows(rank)
        (u_cnt, u_cols, u_val), (i_cnt, i_cols, i_val) = self._share_arrays(layout, rank)
        if csr_factory is None:
            csr_factory = lambda rp, c, v, shape, blocks: CsrMatrix.from_device(rp, c, v, shape, order_blocks=blocks)

        def padded_rowptr(cnt, rows):
            rp = torch.zeros(rows + 1, dtype=torch.int64, device=self.device)
            torch.cumsum(cnt, 0, out=rp[1:cnt.shape[0] + 1])
            rp[cnt.shape[0] + 1:] = rp[cnt.shape[0]]                    # padding rows are empty
            return rp
        u_cols = layout.pad_index_torch(u_cols).to(torch.int32)
        i_cols = layout.pad_index_torch(i_cols).to(torch.int32)
        if layout.fused:
            cnt = torch.zeros(layout.block, dtype=torch.int64, device=self.device)
            cnt[:uhi - ulo] = u_cnt
            cnt[layout.bu:layout.bu + ihi - ilo] = i_cnt
            rp = torch.zeros(layout.block + 1, dtype=torch.int64, device=self.device)
            torch.cumsum(cnt, 0, out=rp[1:])
            return (csr_factory(rp, torch.cat([u_cols, i_cols]), torch.cat([u_val, i_val]), (layout.block, layout.n_pad),
                                [0, layout.bu, layout.block]),)
        return (csr_factory(padded_rowptr(u_cnt, layout.bu), u_cols, u_val, (layout.bu, layout.n_pad), [0, layout.bu]),
                csr_factory(padded_rowptr(i_cnt, layout.bi), i_cols, i_val, (layout.bi, layout.n_pad), [0, layout.bi]))


def check_rows_f64(csr, x, y, rows):
    """max relative error of rows `rows` of y = csr @ x against float64 (by the largest element of each row)."""
    err = 0.0
    rp = csr.rowptr[torch.as_tensor(rows, device=csr.rowptr.device)].tolist()
    rp1 = csr.rowptr[torch.as_tensor(rows, device=csr.rowptr.device) + 1].tolist()
    for r, s, e in zip(rows, rp, rp1):
        ref = (x[csr.col[s:e].long()].double() * csr.val[s:e].double()[:, None]).sum(0)
        err = max(err, float((y[r].double() - ref).abs().max() / (ref.abs().max() + 1e-30)))
    return err

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

    def rank_share(self, layout, rank, xcd_plan=None):
        """(CsrMatrix of the rank's rows [its user block; its item block] with GLOBAL column ids, int64 global row ids)."""
        (ulo, uhi), (ilo, ihi) = layout.user_rows(rank), layout.item_rows(rank)
        dinv = self.inv_sqrt_degree()
        n = self.n
        # user rows: the pair list is sorted by (user, item): the rank's pairs are one contiguous run
        bounds = torch.searchsorted(self.users, torch.tensor([ulo, uhi], device=self.device))
        s, e = int(bounds[0]), int(bounds[1])
        u_rows, u_cols = self.users[s:e] - ulo, self.items[s:e] + self.n_users
        # item rows: pairs whose item lies in the block, sorted by (item, user)
        m = (self.items >= ilo) & (self.items < ihi)
        key = (self.items[m] - ilo) * n + self.users[m]
        key = torch.sort(key).values
        i_rows = torch.div(key, n, rounding_mode='floor')
        i_cols = key - i_rows * n
        del key, m
        nu_l, ni_l = uhi - ulo, ihi - ilo
        counts = torch.cat([self.deg_u[ulo:uhi], self.deg_i[ilo:ihi]])
        rowptr = torch.zeros(nu_l + ni_l + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        col = torch.cat([u_cols, i_cols]).to(torch.int32)
        grow = torch.cat([torch.arange(ulo, uhi, device=self.device), self.n_users + torch.arange(ilo, ihi, device=self.device)])
        row_of = torch.cat([u_rows, nu_l + i_rows])
        val = (dinv[grow][row_of] * 1.0) * dinv[col.long()]              # fl(fl(d_r * 1) * d_c), the reference's order
        del row_of, u_rows, i_rows, u_cols, i_cols
        csr = CsrMatrix.from_device(rowptr, col, val, (nu_l + ni_l, n), order_blocks=[0, nu_l, nu_l + ni_l], xcd_plan=xcd_plan)
        return csr, grow


def check_rows_f64(csr, x, y, rows):
    """max relative error of rows `rows` of y = csr @ x against float64 (by the largest element of each row)."""
    err = 0.0
    rp = csr.rowptr[torch.as_tensor(rows, device=csr.rowptr.device)].tolist()
    rp1 = csr.rowptr[torch.as_tensor(rows, device=csr.rowptr.device) + 1].tolist()
    for r, s, e in zip(rows, rp, rp1):
        ref = (x[csr.col[s:e].long()].double() * csr.val[s:e].double()[:, None]).sum(0)
        err = max(err, float((y[r].double() - ref).abs().max() / (ref.abs().max() + 1e-30)))
    return err

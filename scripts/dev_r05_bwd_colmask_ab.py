"""Developer A/B: the second backward hop of a pruned training step with and without its column mask.

  today:    t1 = A g   [rows: neighbourhood, cols: batch rows; other rows left untouched]
            t2 = g + A t1   [all rows; gathers only the columns in the neighbourhood mask — needed: the rest of t1 is garbage]
  variant:  t1 as above but the other rows written as zeros; t2 = g + A t1 with no column mask (plain gathers)
Amazon-like, d = 64, B = 2048 triplets; interleaved rounds, median."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm, mark_rows, propagate_mean_backward
from scripts.dev_spmm_bench import time_ms

for preset in ('amazon', 'yelp', 'gowalla'):
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021})
    nu, n = ds.n_users, ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ds.n_items)
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan=XCD_PLAN)
    gen = torch.Generator(device='cuda').manual_seed(0)
    ta = torch.from_numpy(ds.train_array).cuda()
    pick = ta[torch.randint(0, ta.shape[0], (2048,), device='cuda', generator=gen)]
    ids = torch.cat([pick[:, 0], nu + pick[:, 1], nu + torch.randint(0, ds.n_items, (2048,), device='cuda', generator=gen)])
    masks = mark_rows(csr, ids)
    g = torch.zeros(n, 64, device='cuda')
    g[masks[0].bool()] = torch.randn(int(masks[0].sum()), 64, device='cuda', generator=gen)
    t1, t2, out = (torch.empty_like(g) for _ in range(3))
    s = 0.25

    def today():
        spmm(csr, g, out=t1, row_mask=masks[1], masked_rows_zero=False, col_mask=masks[2])
        spmm(csr, t1, out=t2, adds=[g], col_mask=masks[3])
        spmm(csr, t2, out=out, adds=[t2], out_scale=s, add_scale=s)

    def variant():
        spmm(csr, g, out=t1, row_mask=masks[1], masked_rows_zero=True, col_mask=masks[2])
        spmm(csr, t1, out=t2, adds=[g])
        spmm(csr, t2, out=out, adds=[t2], out_scale=s, add_scale=s)

    def unmasked():
        spmm(csr, g, out=t1)
        spmm(csr, t1, out=t2, adds=[g])
        spmm(csr, t2, out=out, adds=[t2], out_scale=s, add_scale=s)
    today(); a = out.clone(); variant(); b = out.clone(); unmasked(); c = out.clone()
    res = {'today': [], 'variant': [], 'unmasked': []}
    for rnd in range(5):
        for name, fn in (('today', today), ('variant', variant), ('unmasked', unmasked)):
            res[name].append(time_ms(fn, reps=100, warm=10))
    hop = {}
    hop['hop1_today'] = time_ms(lambda: spmm(csr, g, out=t1, row_mask=masks[1], masked_rows_zero=False, col_mask=masks[2]), 100, 10)
    hop['hop1_zero_fill'] = time_ms(lambda: spmm(csr, g, out=t1, row_mask=masks[1], masked_rows_zero=True, col_mask=masks[2]), 100, 10)
    today()
    hop['hop2_col_mask'] = time_ms(lambda: spmm(csr, t1, out=t2, adds=[g], col_mask=masks[3]), 100, 10)
    variant()
    hop['hop2_plain'] = time_ms(lambda: spmm(csr, t1, out=t2, adds=[g]), 100, 10)
    print(json.dumps({'preset': preset, 'rows_marked': int(masks[0].sum()), 'neighbourhood_rows': int(masks[1].sum()), 'rows': n,
                      'ms': {k: round(sorted(v)[2], 4) for k, v in res.items()}, 'hops_ms': {k: round(v, 4) for k, v in hop.items()},
                      'variant_equals_today': bool(torch.equal(a, b)), 'max_abs_diff_vs_unmasked': float((a - c).abs().max())}), flush=True)

"""Developer (round 5): the dealing order of the XCD plan on the final (two-launch) kernels — where the cut rows' segments sit in
their lists.  Amazon-like d = 64 (and Gowalla- / Yelp-like): list_order x closing_at, one launch and the K = 3 pass, interleaved
rounds, medians."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import propagate_mean, spmm
from scripts.dev_r05_sweeps import time_ms

for preset in os.environ.get('PRESETS', 'amazon,gowalla,yelp').split(','):
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021})
    nu, ni = ds.n_users, ds.n_items
    n = nu + ni
    rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ni)
    x = torch.randn(n, 64, device='cuda') * 0.1
    y = torch.empty_like(x)
    variants = {}
    for at in (0.0, 0.25):
        variants['segments_first closing_at=%.2f' % at] = {'threshold': 112, 'closing_at': at}
    for le, at in ((4, 0.25), (2, 0.25), (1, 0.25), (2, 0.1), (1, 0.1), (1, 0.05)):
        variants['late_every=%d closing_at=%.2f' % (le, at)] = {'threshold': 112, 'closing_at': at, 'late_every': le}
    variants['interleaved'] = {'threshold': 112, 'list_order': 'interleaved'}
    variants['rows_first'] = {'threshold': 112, 'list_order': 'rows_first'}
    mats = {k: CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan=v) for k, v in variants.items()}
    one = {k: [] for k in mats}
    three = {k: [] for k in mats}
    for _ in range(7):
        for k, csr in mats.items():
            one[k].append(time_ms(lambda: spmm(csr, x, out=y), 200, 5))
            three[k].append(time_ms(lambda: propagate_mean(csr, x, 3), 100, 5))
    for k in mats:
        print(json.dumps({'preset': preset, 'order': k, 'one_launch_us': round(float(np.median(one[k])) * 1e3, 2),
                          'pass3_us': round(float(np.median(three[k])) * 1e3, 2)}), flush=True)

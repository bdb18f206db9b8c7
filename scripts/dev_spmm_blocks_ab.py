"""Developer A/B in one process: persistent-grid size of spmm_csr_rows_kernel (blocks of 4 waves per CU)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm, propagate_mean
from scripts.dev_spmm_bench import time_ms

ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
n = ds.n_users + ds.n_items
rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda')
for d in (64,):
    x = torch.randn(n, d, device='cuda') * 0.1
    y = torch.empty_like(x)
    res = {}
    for rnd in range(2):
        for bpc in ('auto', 7, 8, 14, 28, 56, 112, 4096):
            if bpc == 'auto':
                os.environ.pop('IGCN_SPMM_BLOCKS_PER_CU', None)       # library default: measured-residency rule
            else:
                os.environ['IGCN_SPMM_BLOCKS_PER_CU'] = str(bpc)
            ms1 = min(time_ms(lambda: spmm(csr, x, out=y), reps=50) for _ in range(2))
            ms3 = min(time_ms(lambda: propagate_mean(csr, x, 3), reps=30) for _ in range(2))
            res.setdefault(bpc, []).append((round(ms1 * 1e3, 1), round(ms3 * 1e3, 1)))
    print(json.dumps(dict(d=d, us_layer__3layer__by_blocks_per_cu=res)), flush=True)

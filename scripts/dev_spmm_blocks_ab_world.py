"""Developer A/B in one process: grid size of spmm_csr_rows_kernel on the N-GPU bench's per-rank workload
(graph = world x Amazon-like, d = 64 / world columns)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import propagate_mean
from scripts.dev_spmm_bench import time_ms

base = SyntheticDataset.PRESETS['amazon']
for world in [int(w) for w in (sys.argv[1:] or ['2', '4', '8'])]:
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'n_users': base['n_users'] * world, 'n_items': base['n_items'] * world,
                           'n_inter': base['n_inter'] * world, 'seed': 2021, 'device': 'cpu'})
    n = ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda')
    d = 64 // world
    x = torch.randn(n, d, device='cuda') * 0.1
    res = {}
    for rnd in range(2):
        for bpc in ('auto', 6, 7, 8, 14, 28, 56, 112, 4096):
            if bpc == 'auto':
                os.environ.pop('IGCN_SPMM_BLOCKS_PER_CU', None)
            else:
                os.environ['IGCN_SPMM_BLOCKS_PER_CU'] = str(bpc)
            ms3 = min(time_ms(lambda: propagate_mean(csr, x, 3), reps=20) for _ in range(2))
            res.setdefault(bpc, []).append(round(ms3 * 1e3, 1))
    print(json.dumps(dict(world=world, d=d, nnz=int(rowptr[-1]), rows=n, us_3layer_by_blocks_per_cu=res)), flush=True)
    del csr, x
    torch.cuda.empty_cache()

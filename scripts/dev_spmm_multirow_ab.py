"""Developer A/B in one process: one row per wave vs R rows per wave (narrow embeddings) on the N-GPU bench's
per-rank workloads (graph = world x Amazon-like, d = 64 / world columns)."""
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
    res, outs = {}, {}
    for rnd in range(2):
        for mode in ('0', '1'):
            os.environ['IGCN_SPMM_MULTIROW'] = mode
            for bpc in ('auto', '7', '14', '28'):
                if bpc == 'auto':
                    os.environ.pop('IGCN_SPMM_BLOCKS_PER_CU', None)
                else:
                    os.environ['IGCN_SPMM_BLOCKS_PER_CU'] = bpc
                ms = min(time_ms(lambda: propagate_mean(csr, x, 3), reps=20) for _ in range(2))
                res.setdefault('multirow=%s grid=%s' % (mode, bpc), []).append(round(ms * 1e3, 1))
            os.environ.pop('IGCN_SPMM_BLOCKS_PER_CU', None)
            outs[mode] = propagate_mean(csr, x, 3)
    err = float((outs['0'] - outs['1']).abs().max() / outs['0'].abs().max())
    print(json.dumps(dict(world=world, d=d, us_3layer=res, max_rel_diff=err)), flush=True)
    del csr, x, outs
    torch.cuda.empty_cache()

"""Developer: grid size of the SpMM launch under the XCD plan (igcn_set_tuning spmm_blocks_per_cu), Amazon-like, d = 64 / 128."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm
from scripts.dev_spmm_bench import time_ms

for preset in ('amazon', 'gowalla'):
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021})
    nu, n = ds.n_users, ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ds.n_items)
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan=XCD_PLAN)
    for d in (64, 128):
        x = torch.randn(n, d, device='cuda') * 0.1
        y = torch.empty_like(x)
        res = {}
        for rnd in range(3):
            for bpc in (None, 8, 16, 24, 32, 48, 64, 96, 128):
                _lib.set_tuning('spmm_blocks_per_cu', bpc)
                res.setdefault(str(bpc), []).append(time_ms(lambda: spmm(csr, x, out=y), reps=50))
        _lib.set_tuning('spmm_blocks_per_cu', None)
        print(json.dumps({'preset': preset, 'd': d, 'ms_by_blocks_per_cu': {k: round(sorted(v)[1], 4) for k, v in res.items()}}), flush=True)

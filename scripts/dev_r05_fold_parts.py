"""Developer (round 5): what each part of the in-launch fold of cut rows costs — one Amazon-like launch (d = 64, XCD plan: 35 141
segments of 4 133 rows) with "spmm_fold" 0 and 1, for the library IGCN_LIB_PATH names.  The ablation builds
(scripts/dev_build_variant.sh ... spmm.hip -DIGCN_X_FOLD_NOWAIT / -DIGCN_X_FOLD_NOATOMIC) give WRONG results and are only timed:
  storeonly  = agent-scope partial stores, nothing else          noatomic = + the wave waits for its stores
  nowait     = stores + atomic + fold, no wait                    (shipped) = all of it"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm
from scripts.dev_r05_sweeps import time_ms

ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021})
nu, ni = ds.n_users, ds.n_items
n = nu + ni
rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ni)
csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan=XCD_PLAN)
x = torch.randn(n, 64, device='cuda') * 0.1
y = torch.empty_like(x)
res = {'lib': os.environ.get('IGCN_LIB_PATH', 'shipped'), 'n_segments': csr.n_segments}
t = {0: [], 1: []}
for _ in range(7):
    for on in (0, 1):
        _lib.set_tuning('spmm_fold', on)
        t[on].append(time_ms(lambda: spmm(csr, x, out=y), 200, 5))
        csr.partial(64)[csr.n_segments * 64:].zero_()                 # (the ablation builds leave the arrival counters behind)
res['one_launch_ms_fold0'] = round(float(np.median(t[0])), 5)
res['one_launch_ms_fold1'] = round(float(np.median(t[1])), 5)
print(json.dumps(res), flush=True)

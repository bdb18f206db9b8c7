"""Developer (round 5): one Amazon-like A_hat launch (d = 64, XCD plan) and the K = 3 pass, for the library IGCN_LIB_PATH names —
interleave libraries in the shell loop to compare builds of the main SpMM kernel on one box."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import propagate_mean, spmm
from scripts.dev_r05_sweeps import time_ms

ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021})
nu, ni = ds.n_users, ds.n_items
n = nu + ni
rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ni)
csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan=XCD_PLAN)
x = torch.randn(n, 64, device='cuda') * 0.1
y = torch.empty_like(x)
one = [time_ms(lambda: spmm(csr, x, out=y), 300, 10) for _ in range(7)]
three = [time_ms(lambda: propagate_mean(csr, x, 3), 150, 10) for _ in range(7)]
print(json.dumps({'lib': os.environ.get('IGCN_LIB_PATH', 'shipped'), 'one_launch_us': round(float(np.median(one)) * 1e3, 2),
                  'pass3_us': round(float(np.median(three)) * 1e3, 2), 'one_launch_min_us': round(min(one) * 1e3, 2)}), flush=True)

"""A/B in one process: one d=64 SpMM launch vs two d=32 launches on the column halves (smaller gather
working set per pass, index stream read twice)."""
import json, sys
import torch
sys.path.insert(0, '.')
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm
from scripts.dev_spmm_bench import time_ms

ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
n = ds.n_users + ds.n_items
rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda')
x = torch.randn(n, 64, device='cuda') * 0.1
y = torch.empty_like(x)
xa, xb = x[:, :32], x[:, 32:]
ya, yb = y[:, :32], y[:, 32:]
xc = [x[:, 32 * i:32 * i + 32].contiguous() for i in range(2)]
yc = [torch.empty(n, 32, device='cuda') for _ in range(2)]
res = {'full': [], 'split_strided': [], 'split_contig': []}
for _ in range(5):
    res['full'].append(time_ms(lambda: spmm(csr, x, out=y), reps=50))
    res['split_strided'].append(time_ms(lambda: (spmm(csr, xa, out=ya), spmm(csr, xb, out=yb)), reps=50))
    res['split_contig'].append(time_ms(lambda: (spmm(csr, xc[0], out=yc[0]), spmm(csr, xc[1], out=yc[1])), reps=50))
print(json.dumps({k: [round(v, 4) for v in vs] for k, vs in res.items()}))

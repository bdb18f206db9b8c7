"""Developer (round 5): every kind of SpMM launch of an IGCN training step on the Yelp-like split, timed back to back, for the library
IGCN_LIB_PATH names — to find which launch the round-5 kernel changes made slower (the captured step went 0.637 -> 0.686 ms)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from igcn_cf_amd import config as cfg
from igcn_cf_amd import ops
from igcn_cf_amd.dataset import get_dataset
from igcn_cf_amd.model import get_model

dev = torch.device('cuda', 0)
ds_cfg, m_cfg, _ = cfg.get_synthetic_config(dev, 'yelp')[2]
ds = get_dataset(ds_cfg)
torch.manual_seed(2021)
ig = get_model(dict(m_cfg, embedding_size=64, n_layers=3), ds)
ig.eval()
if ig._feat_scale is None:
    ig.update_feat_mat()
F, T, scale, A = ig.feat_mat, ig.embedding.weight.detach(), ig._feat_scale, ig.norm_adj
Ft = F.transposed_view()
n = A.shape[0]
g = torch.Generator(device=dev).manual_seed(1)
x0 = ops.spmm(F, T, row_scale=scale)
grad = torch.randn(n, 64, device=dev, generator=g)
y = torch.empty_like(x0)
yt = torch.empty(Ft.shape[0], 64, device=dev)
batch = torch.randint(0, n, (3 * 2048,), device=dev, generator=g)
m1, m2, b1, b2 = ops.mark_rows(A, batch)
res = {'lib': os.environ.get('IGCN_LIB_PATH', 'shipped'), 'F': [F.shape[0], F.nnz, F.n_long, F.n_segments], 'Ft': [Ft.shape[0], Ft.nnz, Ft.n_long, Ft.n_segments],
       'A': [A.shape[0], A.nnz, A.n_long, A.n_segments]}
cases = {
    'F_T_eval': lambda: ops.spmm(F, T, out=y, row_scale=scale),
    'F_T_dropout': lambda: ops.spmm(F, T, out=y, row_scale=scale, keep_prob=0.7, seed=5),
    'Ft_backward_dropout': lambda: ops.spmm(Ft, grad, out=yt, col_scale=scale, keep_prob=0.7, seed=5),
    'A_hat': lambda: ops.spmm(A, x0, out=y),
    'A_hat_with_adds': lambda: ops.spmm(A, x0, out=y, adds=[x0, grad, x0], out_scale=0.25, add_scale=0.25),
}
if m1 is not None:
    cases['A_hat_rows_of_the_batch'] = lambda: ops.spmm(A, x0, out=y, row_mask=m1, masked_rows_zero=True)
    cases['A_hat_rows_and_neighbours'] = lambda: ops.spmm(A, x0, out=y, row_mask=m2, masked_rows_zero=False)
    cases['A_hat_backward_hop_col_mask'] = lambda: ops.spmm(A, grad, out=y, row_mask=m2, masked_rows_zero=True, col_mask=b1)
for name, fn in cases.items():
    res[name] = [round(bench.time_ms(fn, 300, 20) * 1e3, 2) for _ in range(3)]
print(json.dumps(res), flush=True)

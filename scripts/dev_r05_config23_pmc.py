"""The launches whose counters profiles/pmc_traffic_config23.json holds (run under rocprofv3 by scripts/profile_config23.sh):
BASELINE config 2 (LightGCN on the Gowalla-like split) and config 3 (IGCN on the Yelp-like split), d = 64 — of each graph one plain
A_hat launch and the two launches of the pass that carry an epilogue addend (ops.mean_plan at K = 3: U = X_0 + A X_1, then the
result s (U + A U)), and the RECTANGULAR feature launch X0 = F T of IGCN.inductive_rep_layer
(model.py:423-432) in eval mode and with the config's edge dropout 0.3.  Each variant N_LAUNCH times IN THIS ORDER with no other
dispatch of the main SpMM kernels in between, so that the summariser can tell them apart by dispatch order.  Prints one JSON line
with the sizes, the algorithmic bytes and the HIP-event times of the same launches."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import config as cfg
from igcn_cf_amd import ops
from igcn_cf_amd.dataset import get_dataset
from igcn_cf_amd.model import get_model

N_LAUNCH = 4                                    # per variant; the summariser drops the first (cold) one
dev = torch.device('cuda', 0)
d, K = 64, 3
variants = []                                   # (name, launch function, rows, nnz, bytes per index entry)
keep = []

ds_cfg, m_cfg, _ = cfg.get_synthetic_config(dev, 'gowalla')[1]
ds = get_dataset(ds_cfg)
torch.manual_seed(2021)
lg = get_model(dict(m_cfg, embedding_size=d, n_layers=K), ds)
A = lg.norm_adj
x = lg.embedding.weight.detach()
assert ops.mean_plan(K) == [None, 0, 2]
layers = [x, ops.spmm(A, x), None]
layers[2] = ops.spmm(A, layers[1], adds=[x])
y = torch.empty_like(x)
variants.append(('gowalla_A_hat', lambda A=A, x=x, y=y: ops.spmm(A, x, out=y), A.shape[0], A.nnz, 8, 0))
variants.append(('gowalla_A_hat_launch_with_addend', lambda A=A, l=layers, y=y: ops.spmm(A, l[1], out=y, adds=[l[0]]), A.shape[0], A.nnz, 8, 1))
variants.append(('gowalla_A_hat_last_launch', lambda A=A, l=layers, y=y: ops.spmm(A, l[2], out=y, adds=[l[2]], out_scale=0.25, add_scale=0.25),
                 A.shape[0], A.nnz, 8, 1))
keep += [lg, layers, y]

ds_cfg, m_cfg, _ = cfg.get_synthetic_config(dev, 'yelp')[2]
ds3 = get_dataset(ds_cfg)
torch.manual_seed(2021)
ig = get_model(dict(m_cfg, embedding_size=d, n_layers=K), ds3)
ig.eval()
if ig._feat_scale is None:
    ig.update_feat_mat()
F, T, scale = ig.feat_mat, ig.embedding.weight.detach(), ig._feat_scale
x0 = ops.spmm(F, T, row_scale=scale)
y0 = torch.empty_like(x0)
variants.append(('yelp_F_T_eval', lambda: ops.spmm(F, T, out=y0, row_scale=scale), F.shape[0], F.nnz, 4, 0))
variants.append(('yelp_F_T_dropout_0.3', lambda: ops.spmm(F, T, out=y0, row_scale=scale, keep_prob=0.7, seed=12345), F.shape[0], F.nnz, 4, 0))
A3 = ig.norm_adj
l3 = [x0, ops.spmm(A3, x0), None]
l3[2] = ops.spmm(A3, l3[1], adds=[x0])
y3 = torch.empty_like(x0)
variants.append(('yelp_A_hat', lambda: ops.spmm(A3, x0, out=y3), A3.shape[0], A3.nnz, 8, 0))
variants.append(('yelp_A_hat_launch_with_addend', lambda: ops.spmm(A3, l3[1], out=y3, adds=[l3[0]]), A3.shape[0], A3.nnz, 8, 1))
variants.append(('yelp_A_hat_last_launch', lambda: ops.spmm(A3, l3[2], out=y3, adds=[l3[2]], out_scale=0.25, add_scale=0.25), A3.shape[0], A3.nnz, 8, 1))

out = {'d': d, 'n_layers': K, 'n_launch': N_LAUNCH, 'order': [v[0] for v in variants], 'launches': {},
       'gowalla': {'users': ds.n_users, 'items': ds.n_items, 'nnz_A_hat': A.nnz},
       'yelp': {'users': ds3.n_users, 'items': ds3.n_items, 'nnz_A_hat': A3.nnz, 'nnz_F': F.nnz, 'F_shape': list(F.shape), 'templates': int(T.shape[0])}}
for name, fn, rows, nnz, idx_bytes, n_adds in variants:
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(N_LAUNCH + 1)]
    e[0].record()
    for j in range(N_LAUNCH):
        fn()
        e[j + 1].record()
    torch.cuda.synchronize()
    out['launches'][name] = {'rows': rows, 'nnz': nnz, 'ms': [e[j].elapsed_time(e[j + 1]) for j in range(N_LAUNCH)],
                             # per stored nonzero its index bytes (col + val; F has no value array: its values are a per-row
                             # scale) + one gathered source row; per output row the row + its pointer (+ the mean's addends where fused)
                             'algorithmic_bytes': nnz * (idx_bytes + 4 * d) + rows * (4 * d + 4) + n_adds * rows * 4 * d}
print(json.dumps(out))

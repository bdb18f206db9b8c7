"""A/B in one process: SpMM on the Amazon-like graph with the original node order vs nodes relabelled
(a) by descending degree inside the user / item blocks, (b) users by the smallest hot item they touch."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm
from scripts.dev_spmm_bench import time_ms

ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
nu, ni = ds.n_users, ds.n_items
ta = ds.train_array
deg_u = np.bincount(ta[:, 0], minlength=nu)
deg_i = np.bincount(ta[:, 1], minlength=ni)


def relabel(perm_u, perm_i):            # perm[new] = old
    inv_u = np.empty(nu, dtype=np.int64); inv_u[perm_u] = np.arange(nu)
    inv_i = np.empty(ni, dtype=np.int64); inv_i[perm_i] = np.arange(ni)
    return np.stack([inv_u[ta[:, 0]], inv_i[ta[:, 1]]], axis=1)


variants = {'original': ta}
pu, pi = np.argsort(-deg_u, kind='stable'), np.argsort(-deg_i, kind='stable')
variants['degree_sorted'] = relabel(pu, pi)
rank_i = np.empty(ni, dtype=np.int64); rank_i[pi] = np.arange(ni)
first_hot = np.full(nu, ni, dtype=np.int64)
np.minimum.at(first_hot, ta[:, 0], rank_i[ta[:, 1]])
variants['items_by_degree_users_by_hottest_item'] = relabel(np.argsort(first_hot, kind='stable'), pi)
try:                                     # reverse Cuthill-McKee on the bipartite graph (bandwidth-reducing order)
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    a = sp.coo_matrix((np.ones(len(ta)), (ta[:, 0], ta[:, 1] + nu)), shape=(nu + ni, nu + ni)).tocsr()
    perm = reverse_cuthill_mckee((a + a.T).tocsr(), symmetric_mode=True)
    variants['reverse_cuthill_mckee'] = relabel(perm[perm < nu], perm[perm >= nu] - nu)
except Exception as e:
    print('rcm skipped', repr(e))
variants['random'] = relabel(np.random.default_rng(0).permutation(nu), np.random.default_rng(1).permutation(ni))

mats = {}
for name, arr in variants.items():
    rowptr, col, val = normalized_adjacency_host(arr, nu, ni)
    mats[name] = CsrMatrix(rowptr, col, val, (nu + ni, nu + ni), 'cuda', order_blocks=[0, nu, nu + ni])
x = torch.randn(nu + ni, 64, device='cuda') * 0.1
y = torch.empty_like(x)
res = {k: [] for k in mats}
for rnd in range(5):
    for name, csr in mats.items():
        res[name].append(time_ms(lambda: spmm(csr, x, out=y), reps=50))
print(json.dumps({k: round(sorted(v)[2], 4) for k, v in res.items()}))

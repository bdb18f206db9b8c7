"""Developer A/B: how the layer mean of get_rep is folded into the K launches.  Today: the last launch adds X_0..X_{K-1}
(three extra row reads in ONE launch, four [n, d] tables alive).  Horner: Y = s (X_0 + A (X_0 + A (X_0 + A X_0))), one extra
row read per launch, always of X_0, three tables alive (the result overwrites the first intermediate)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import normalized_adjacency_device
from igcn_cf_amd.ops import propagate_mean, spmm

dev = torch.device('cuda')
for preset, d in (('amazon', 64), ('amazon', 128), ('gowalla', 64), ('yelp', 64)):
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset})
    n = ds.n_users + ds.n_items
    csr = normalized_adjacency_device(ds.train_array, ds.n_users, ds.n_items, dev)
    x0 = torch.randn(n, d, device=dev) * 0.1
    K, s = 3, 0.25
    t1, t2 = torch.empty_like(x0), torch.empty_like(x0)

    def horner():
        spmm(csr, x0, adds=[x0], out_scale=1.0, add_scale=1.0, out=t1)
        spmm(csr, t1, adds=[x0], out_scale=1.0, add_scale=1.0, out=t2)
        return spmm(csr, t2, adds=[x0], out_scale=s, add_scale=s, out=t1)

    def plain_no_mean():                                         # what the three launches cost without any add
        spmm(csr, x0, out=t1)
        spmm(csr, t1, out=t2)
        return spmm(csr, t2, out=t1)
    ref = propagate_mean(csr, x0, K)
    got = horner().clone()
    want = ref.double()
    # float64 truth of the mean
    coo = csr.to_torch_coo().double()
    xs, acc = x0.double(), x0.double().clone()
    for _ in range(K):
        xs = torch.sparse.mm(coo, xs)
        acc += xs
    acc *= s
    rec = dict(preset=preset, d=d,
               today_us=round(min(bench.time_ms(lambda: propagate_mean(csr, x0, K), 100, 10) for _ in range(3)) * 1e3, 1),
               horner_us=round(min(bench.time_ms(horner, 100, 10) for _ in range(3)) * 1e3, 1),
               no_mean_us=round(min(bench.time_ms(plain_no_mean, 100, 10) for _ in range(3)) * 1e3, 1),
               today_max_err_vs_f64=float((ref.double() - acc).abs().max()), horner_max_err_vs_f64=float((got.double() - acc).abs().max()),
               scale=float(acc.abs().max()))
    print(json.dumps(rec), flush=True)
    del csr, x0, t1, t2, coo, xs, acc
    torch.cuda.empty_cache()

"""XCD plan: a short row goes to the list that holds most of its columns.  Round 6 A/B: prefer the list that holds most of its COLD
columns instead (cfg 'cold_rows' = H: the H most-gathered operand rows of a phase count as hot — they sit in every XCD's L2 whoever
gathers them).  One A_hat launch, d = 64, same process, interleaved rounds, HIP events; results must be bit-equal to rounding."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import ops
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host

dev = torch.device('cuda', 0)
d = 64
for preset in (sys.argv[1:] or ['amazon', 'yelp', 'gowalla']):
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021, 'device': dev})
    n = ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
    g = torch.Generator(device='cpu').manual_seed(2021)
    x = (torch.randn(n, d, generator=g) * 0.1).to(dev)
    variants = {}
    for H in (0, 2048, 8192, 16384, 32768):
        csr = CsrMatrix(rowptr, col, val, (n, n), dev, order_blocks=[0, ds.n_users, n], xcd_plan={'threshold': 112, 'cold_rows': H})
        variants[H] = (csr, torch.empty_like(x))
    ref = ops.spmm(variants[0][0], x).clone()
    times = {H: [] for H in variants}
    for rnd in range(5):
        for H, (csr, y) in variants.items():
            for _ in range(30):
                ops.spmm(csr, x, out=y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                ops.spmm(csr, x, out=y)
            e1.record()
            torch.cuda.synchronize()
            times[H].append(e0.elapsed_time(e1) / 300 * 1e3)
    rec = {'preset': preset, 'nnz': int(rowptr[-1])}
    for H, (csr, y) in variants.items():
        rec['cold_rows_%d_us' % H] = round(sorted(times[H])[2], 2)
        rec['cold_rows_%d_err' % H] = float((y - ref).abs().max() / ref.abs().max())
    print(json.dumps(rec), flush=True)

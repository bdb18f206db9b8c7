"""Developer (round 5): how long does an idle MI355X need to run the headline pass at full speed?  Per-10-pass GPU times (HIP events)
of 600 passes behind (a) two seconds of idle, (b) the stream + gather probes of bench.py (what the driver's run now sees)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import propagate_mean

dev = torch.device('cuda', 0)
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021, 'device': dev})
nu, ni = ds.n_users, ds.n_items
n = nu + ni
rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ni)
csr = CsrMatrix(rowptr, col, val, (n, n), dev, order_blocks=[0, nu, n], xcd_plan=XCD_PLAN)
x = torch.randn(n, 64, device=dev) * 0.1
propagate_mean(csr, x, 3)
torch.cuda.synchronize()
for label in ('after 2 s idle', 'after the probes'):
    if label == 'after 2 s idle':
        time.sleep(2.0)
    else:
        time.sleep(2.0)
        bench.measured_stream(dev)
        bench.gather_roof(dev, csr.col, csr.val, x, n, 64)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    ev[0].record()
    for b in range(60):
        for _ in range(10):
            propagate_mean(csr, x, 3)
        ev[b + 1].record()
    torch.cuda.synchronize()
    ms = [round(ev[b].elapsed_time(ev[b + 1]) / 10, 4) for b in range(60)]
    print(json.dumps({'start': label, 'ms_per_pass_in_blocks_of_10': ms}), flush=True)

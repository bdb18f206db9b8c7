"""Developer A/B of two builds of the library (same ABI) in ONE process, alternating: 3-layer propagation on the
N-GPU bench's per-rank workloads.   python scripts/dev_spmm_ab_libs.py libA.so libB.so [worlds...]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import igcn_cf_amd._lib as _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import propagate_mean
from scripts.dev_spmm_bench import time_ms

paths = [a for a in sys.argv[1:] if a.endswith('.so')]
worlds = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8]
libs = {os.path.basename(p): C.CDLL(os.path.abspath(p)) for p in paths}
_lib._handle, _lib._bound = next(iter(libs.values())), {}
base = SyntheticDataset.PRESETS['amazon']
for world in worlds:
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'n_users': base['n_users'] * world, 'n_items': base['n_items'] * world,
                           'n_inter': base['n_inter'] * world, 'seed': 2021, 'device': 'cpu'})
    n = ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda')
    d = 64 // world
    x = torch.randn(n, d, device='cuda') * 0.1
    res, outs = {}, {}
    for rnd in range(3):
        for name, handle in libs.items():
            _lib._handle, _lib._bound = handle, {}
            res.setdefault(name, []).append(round(min(time_ms(lambda: propagate_mean(csr, x, 3), reps=20) for _ in range(2)) * 1e3, 1))
            outs[name] = propagate_mean(csr, x, 3)
    names = list(libs)
    print(json.dumps(dict(world=world, d=d, us_3layer=res, identical=all(torch.equal(outs[names[0]], outs[m]) for m in names[1:]))), flush=True)
    del csr, x, outs
    torch.cuda.empty_cache()

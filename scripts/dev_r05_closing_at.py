"""Developer (round 5): where in a list should the CLOSING segments of the cut rows sit?  Amazon-like, d = 64, XCD plan; closing_at =
fraction into the phase's rows (0 = right behind the phase's other segments); one launch and the K = 3 pass, "spmm_fold" 1 against 0,
same process, interleaved rounds, medians; polls counted by nothing — a closing segment that comes too early simply spins."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import propagate_mean, spmm
from scripts.dev_r05_sweeps import time_ms

for preset in os.environ.get('PRESETS', 'amazon,gowalla,yelp').split(','):
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021})
    nu, ni = ds.n_users, ds.n_items
    n = nu + ni
    rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ni)
    x = torch.randn(n, 64, device='cuda') * 0.1
    y = torch.empty_like(x)
    ref = None
    for at in (0.0, 0.05, 0.1, 0.25, 0.5):
        csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan={'threshold': 112, 'closing_at': at})
        one, three = {0: [], 1: []}, {0: [], 1: []}
        for _ in range(7):
            for on in (0, 1):
                _lib.set_tuning('spmm_fold', on)
                one[on].append(time_ms(lambda: spmm(csr, x, out=y), 200, 5))
                three[on].append(time_ms(lambda: propagate_mean(csr, x, 3), 100, 5))
        _lib.set_tuning('spmm_fold', 1)
        out = spmm(csr, x).clone()
        if ref is None:
            _lib.set_tuning('spmm_fold', 0)
            ref = spmm(csr, x).clone()
        res = {'preset': preset, 'closing_at': at, 'bit_equal_to_two_launch_form': bool(torch.equal(out, ref))}
        for on in (0, 1):
            res['one_launch_ms_fold%d' % on] = round(float(np.median(one[on])), 5)
            res['pass3_ms_fold%d' % on] = round(float(np.median(three[on])), 5)
        res['pass3_delta_pct'] = round(100 * (res['pass3_ms_fold1'] / res['pass3_ms_fold0'] - 1), 2)
        print(json.dumps(res), flush=True)
_lib.set_tuning('spmm_fold', None)

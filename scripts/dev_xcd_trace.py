"""Developer tool: per-XCD finish times of one SpMM launch under the XCD plan (trace build, -DIGCN_SPMM_TRACE):
is the static split of the work over the eight lists balanced in TIME, and do workgroups b, b + 8 really share an XCD?"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.dev_spmm_trace import TRACE_LIB, build


def main():
    build()
    import torch
    import igcn_cf_amd._lib as _lib
    _lib.LIB_PATH = TRACE_LIB
    from igcn_cf_amd import ops
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
    wt = _lib.handle().igcn_debug_spmm_wave_times
    wt.restype, wt.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_int]
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021})
    nu, n = ds.n_users, ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ds.n_items)
    x = torch.randn(n, 64, device='cuda') * 0.1
    plans = {'legacy': None}
    for T in (96, 128):
        for rc in (4, 16):
            plans['xcd_T%d_rc%d' % (T, rc)] = {'threshold': T, 'row_cost': rc}
    for name, plan in plans.items():
        csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan=plan)
        for _ in range(3):
            ops.spmm(csr, x)
        torch.cuda.synchronize()
        nw = 131072
        buf = (C.c_uint64 * (6 * nw))()
        ops.spmm(csr, x)
        torch.cuda.synchronize()
        wt(buf, nw)
        w = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 6).astype(np.int64)
        used = (w[:, 1] > 0) & (w[:, 0] >= w[:, 0].max() - 50000)       # entries of THIS launch (the buffer is never cleared)
        idx = np.flatnonzero(used)
        w = w[used]
        t0 = w[:, 0].min()
        b, e = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
        res = idx // 4 % 8                                      # workgroup index modulo 8
        xcc = (w[:, 2] >> 32) & 15
        agree = {int(r): np.bincount(xcc[res == r], minlength=8).tolist() for r in range(8)}
        out = {'plan': name, 'waves_traced': int(used.sum()), 'kernel_us': round(float(e.max()), 1),
               'end_us_by_list': [round(float(e[res == r].max()), 1) for r in range(8)],
               'median_end_us_by_list': [round(float(np.median(e[res == r])), 1) for r in range(8)],
               'nnz_by_list': [int(w[res == r, 4].sum()) for r in range(8)],
               'rows_by_list': [int(w[res == r, 3].sum()) for r in range(8)],
               'begin_us_quantiles': [round(float(v), 1) for v in np.quantile(b, [0.1, 0.5, 0.9, 1.0])],
               'busy_us_quantiles': [round(float(v), 2) for v in np.quantile(e - b, [0.1, 0.5, 0.9, 0.99, 1.0])],
               'residue_equals_xcc_id': bool(all(agree[r][r] == sum(agree[r]) for r in range(8)))}
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()

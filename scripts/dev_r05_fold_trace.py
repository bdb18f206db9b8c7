"""Developer (round 5): per-wave timeline of ONE Amazon-like SpMM launch (d = 64, XCD plan) with the cut rows added up inside the
launch ("spmm_fold" 1) and by the second kernel (0) — trace build of spmm.hip (-DIGCN_SPMM_TRACE: every wave's begin / end in 100 MHz
ticks).  Waves are classed by what the dealing order handed them: closing segments, other segments, rows.
    bash scripts/dev_build_variant.sh trace spmm.hip -DIGCN_SPMM_TRACE && IGCN_LIB_PATH=ab_libs/libigcn_hip_trace.so python scripts/dev_r05_fold_trace.py"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm

wt = _lib.handle().igcn_debug_spmm_wave_times
wt.restype, wt.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_int]
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021})
nu, ni = ds.n_users, ds.n_items
n = nu + ni
rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ni)
csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan=XCD_PLAN)
x = torch.randn(n, 64, device='cuda') * 0.1
y = torch.empty_like(x)
order = csr.row_order.cpu().numpy().astype(np.int64)
xoff = csr.xcd_off.cpu().numpy()
seg = csr.segments.view(torch.int32).view(-1, 6).cpu().numpy()
closing = seg[:, 5] < 0
kind_of = np.zeros(order.shape[0], dtype=np.int64)                      # 0 row, 1 segment, 2 closing segment
is_seg = order >= n
kind_of[is_seg] = 1 + closing[order[is_seg] - n]
nw = 131072
for on in (1, 0):
    _lib.set_tuning('spmm_fold', on)
    for _ in range(3):
        spmm(csr, x, out=y)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * (6 * nw))()
    spmm(csr, x, out=y)
    torch.cuda.synchronize()
    wt(buf, nw)
    w = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 6).astype(np.int64)
    idx = np.flatnonzero(w[:, 1] > 0)
    w = w[idx]
    t0 = w[:, 0].min()
    b, e = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
    # entries of each wave: list x = block & 7, wave_x = (block >> 3) * 4 + wave in block, entries first .. first + 1 (R = 2), one iteration
    block, wib = idx // 4, idx % 4
    xl = block & 7
    first = xoff[xl] + ((block >> 3) * 4 + wib) * 2
    k0 = np.where(first < xoff[xl + 1], kind_of[np.minimum(first, order.shape[0] - 1)], -1)
    k1 = np.where(first + 1 < xoff[xl + 1], kind_of[np.minimum(first + 1, order.shape[0] - 1)], -1)
    cls = np.maximum(k0, k1)
    res = {'spmm_fold': on, 'waves': int(len(w)), 'kernel_us': round(float(e.max()), 2)}
    for name, c in (('rows', 0), ('segments', 1), ('closing', 2)):
        m = cls == c
        if m.any():
            d = (e - b)[m]
            res[name] = {'waves': int(m.sum()), 'busy_us_quantiles_0_50_90_99_100': [round(float(v), 2) for v in np.quantile(d, [0, 0.5, 0.9, 0.99, 1.0])],
                         'begin_us_quantiles_0_50_100': [round(float(v), 1) for v in np.quantile(b[m], [0, 0.5, 1.0])],
                         'wave_time_share': round(float(d.sum() / (e - b).sum()), 4)}
    res['total_wave_time_us'] = round(float((e - b).sum()), 0)
    print(json.dumps(res), flush=True)

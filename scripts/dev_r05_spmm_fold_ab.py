"""Same-process A/B (round 5): cut rows folded INSIDE the SpMM launch ("spmm_fold" 1: the wave that delivers a row's last partial sum
adds the row up) against the two-launch form ("spmm_fold" 0: spmm_long_rows_reduce_kernel behind every launch).  Per graph and width:
one launch and the K = 3 pass with the layer mean (ops.propagate_mean: LightGCN.get_rep), interleaved rounds, medians; the outputs of
the two forms must be BIT-EQUAL (both add a row's partial sums in slot order), with row masks and masked_rows_zero too.
    python scripts/dev_r05_spmm_fold_ab.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import propagate_mean, spmm


def time_ms(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fold(on):
    _lib.set_tuning('spmm_fold', 1 if on else 0)


for preset, d, plan in (('amazon', 64, XCD_PLAN), ('amazon', 64, None), ('amazon', 128, XCD_PLAN), ('gowalla', 64, XCD_PLAN), ('yelp', 64, XCD_PLAN),
                        ('amazon', 32, None), ('amazon', 16, XCD_PLAN)):
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021})
    nu, ni = ds.n_users, ds.n_items
    n = nu + ni
    rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ni)
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan=plan)
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(n, d, device='cuda', generator=g) * 0.1
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    # bit-equality: plain launch, pass with the mean, and a masked launch that must zero the rows it skips
    fold(False); spmm(csr, x, out=y0); p0 = propagate_mean(csr, x, 3)
    fold(True); spmm(csr, x, out=y1); p1 = propagate_mean(csr, x, 3)
    same = bool(torch.equal(y0, y1)) and bool(torch.equal(p0, p1))
    mask = (torch.rand(n, device='cuda', generator=g) < 0.3).to(torch.uint8)
    m0, m1 = torch.full_like(x, 7.0), torch.full_like(x, 7.0)
    fold(False); spmm(csr, x, out=m0, row_mask=mask, masked_rows_zero=True)
    fold(True); spmm(csr, x, out=m1, row_mask=mask, masked_rows_zero=True)
    same_masked = bool(torch.equal(m0, m1))
    # ... and repeated launches leave the arrival counters at zero (a second folded launch gives the same bits)
    fold(True); spmm(csr, x, out=y1); again = bool(torch.equal(y0, y1))
    ref = torch.sparse.mm(csr.to_torch_coo().double(), x.double()) if n < 150000 else None
    res = {'preset': preset, 'd': d, 'plan': 'xcd' if plan else 'plain', 'nnz': int(rowptr[-1]), 'n_long': csr.n_long, 'n_segments': csr.n_segments,
           'bit_equal': same, 'bit_equal_masked_zeroing': same_masked, 'bit_equal_second_folded_launch': again}
    if ref is not None:
        res['rel_err_vs_f64'] = float((y1.double() - ref).abs().max() / ref.abs().max())
    one = {0: [], 1: []}
    three = {0: [], 1: []}
    for _ in range(7):
        for on in (0, 1):
            fold(bool(on))
            one[on].append(time_ms(lambda: spmm(csr, x, out=y1), 200))
            three[on].append(time_ms(lambda: propagate_mean(csr, x, 3), 100))
    for on in (0, 1):
        res['one_launch_ms_fold%d' % on] = round(float(np.median(one[on])), 5)
        res['pass3_ms_fold%d' % on] = round(float(np.median(three[on])), 5)
    res['one_launch_delta_pct'] = round(100 * (res['one_launch_ms_fold1'] / res['one_launch_ms_fold0'] - 1), 2)
    res['pass3_delta_pct'] = round(100 * (res['pass3_ms_fold1'] / res['pass3_ms_fold0'] - 1), 2)
    print(json.dumps(res), flush=True)
_lib.set_tuning('spmm_fold', None)

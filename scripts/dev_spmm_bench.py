"""Developer micro-benchmark of the CSR SpMM kernel (not the driver's bench.py)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm, propagate_mean


def time_ms(fn, reps=20, warm=3):
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


def main():
    out = {}
    # empirical copy peak
    a = torch.empty(1 << 28, dtype=torch.float32, device='cuda')  # 1 GiB
    b = torch.empty_like(a)
    ms = time_ms(lambda: b.copy_(a), reps=10)
    out['copy_GBps'] = 2 * a.numel() * 4 / ms / 1e6
    del a, b
    for preset, d in (('gowalla', 64), ('yelp', 64), ('amazon', 64), ('amazon', 128)):
        for zq in (150.0, 0.0):
            ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'zipf_q': zq})
            n = ds.n_users + ds.n_items
            rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
            for lt, sl in ((1024, 512), (256, 256), (4096, 1024)):
                csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', long_threshold=lt, segment_len=sl)
                x = torch.randn(n, d, device='cuda') * 0.1
                y = torch.empty_like(x)
                ms1 = time_ms(lambda: spmm(csr, x, out=y))
                ms3 = time_ms(lambda: propagate_mean(csr, x, 3))
                nnz = csr.nnz
                balg = nnz * (8 + 4 * d) + n * (4 * d + 4)
                rec = dict(preset=preset, d=d, zipf_q=zq, n=n, nnz=nnz, long_threshold=lt, segment_len=sl,
                           n_long=csr.n_long, n_segments=csr.n_segments, max_deg=int(np.diff(rowptr).max()),
                           ms_layer=ms1, ms_3layer=ms3, gedges_per_s=nnz / ms1 / 1e6,
                           alg_GBps=balg / ms1 / 1e6, frac_8TBps=balg / ms1 / 1e6 / 8000)
                print(json.dumps(rec), flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

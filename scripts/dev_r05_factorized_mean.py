"""Developer A/B: the K = 3 layer mean as s (I + A)(I + A^2) X_0 — two epilogue addends instead of three.

  today:       X1 = A X0; X2 = A X1; out = s (X0 + X1 + X2 + A X2)          addends read: X0, X1, X2 (last launch)
  factorised:  X1 = A X0; U = X0 + A X1; out = s (U + A U)                   addends read: X0 (2nd launch), U (last launch)

Same three products, one table read less per pass; the sums are associated differently (fp32 rounding only).
Prints both times interleaved (median of rounds) and each form's error against a float64 chain."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm, propagate_mean
from scripts.dev_spmm_bench import time_ms


def factorised(csr, x0, bufs):
    x1, u, y = bufs
    spmm(csr, x0, out=x1)
    spmm(csr, x1, out=u, adds=[x0], out_scale=1.0, add_scale=1.0)
    spmm(csr, u, out=y, adds=[u], out_scale=0.25, add_scale=0.25)
    return y


def main():
    for preset, d in (('amazon', 64), ('gowalla', 64), ('yelp', 64), ('amazon', 128)):
        ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021})
        nu, n = ds.n_users, ds.n_users + ds.n_items
        rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ds.n_items)
        csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan=XCD_PLAN)
        x0 = torch.randn(n, d, device='cuda') * 0.1
        bufs = [torch.empty_like(x0) for _ in range(3)]
        a = propagate_mean(csr, x0, 3)
        b = factorised(csr, x0, bufs).clone()
        m = torch.sparse_csr_tensor(torch.from_numpy(rowptr), torch.from_numpy(col.astype(np.int64)), torch.from_numpy(val).double(),
                                    size=(n, n)).cuda()
        x = x0.double()
        acc = x.clone()
        for _ in range(3):
            x = m @ x
            acc += x
        ref = acc / 4
        scale = float(ref.abs().max())
        res = {'today': [], 'factorised': []}
        for rnd in range(5):
            res['today'].append(time_ms(lambda: propagate_mean(csr, x0, 3), reps=100, warm=20))
            res['factorised'].append(time_ms(lambda: factorised(csr, x0, bufs), reps=100, warm=20))
        print(json.dumps({'preset': preset, 'd': d, 'ms_today': round(sorted(res['today'])[2], 4),
                          'ms_factorised': round(sorted(res['factorised'])[2], 4),
                          'max_abs_err_today_over_max': float((a.double() - ref).abs().max()) / scale,
                          'max_abs_err_factorised_over_max': float((b.double() - ref).abs().max()) / scale,
                          'max_rel_rowwise_today': float(((a.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-30)).max()),
                          'max_rel_rowwise_factorised': float(((b.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-30)).max())}),
              flush=True)


if __name__ == '__main__':
    main()

"""Developer: the bf16x3 evaluation path next to the exact fp32 one — agreement and time."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import _csr_to_device, _merge_sorted_csr
from scripts.dev_spmm_bench import time_ms

# exactness on integer-valued embeddings (every plane beyond the first is zero)
rng = np.random.default_rng(0)
U = torch.from_numpy(rng.integers(-4, 5, size=(700, 64)).astype(np.float32)).cuda()
I = torch.from_numpy(rng.integers(-4, 5, size=(5000, 64)).astype(np.float32)).cuda()
a, b = score_topk(U, I, 20), score_topk(U, I, 20, precision='bf16x3')
print(json.dumps(dict(integer_case_identical=bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])))))

ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
g = torch.Generator(device='cuda').manual_seed(0)
U = torch.randn(ds.n_users, 64, device='cuda', generator=g) * 0.1
I = torch.randn(ds.n_items, 64, device='cuda', generator=g) * 0.1
excl = _merge_sorted_csr(ds.csr('train'), ds.csr('val'))
rp, cl = _csr_to_device(excl[0], excl[1], 'cuda')
users = torch.arange(ds.n_users, device='cuda')
kw = dict(user_ids=users, excl_rowptr=rp, excl_col=cl)
i0, v0 = score_topk(U, I, 20, **kw)
i1, v1 = score_topk(U, I, 20, precision='bf16x3', **kw)
same_rows = (i0 == i1).all(dim=1)
sets_equal = torch.tensor([set(x.tolist()) == set(y.tolist()) for x, y in zip(i0[~same_rows][:2000].cpu(), i1[~same_rows][:2000].cpu())])
ref = (U[:2048].double() @ I.double().T)
r0 = torch.gather(ref, 1, i0[:2048])
print(json.dumps(dict(rows_identical=float(same_rows.float().mean()), differing_rows=int((~same_rows).sum()),
                      differing_rows_with_equal_sets_of_first_2000=float(sets_equal.float().mean()) if len(sets_equal) else None,
                      max_rel_err_fp32_vs_f64=float(((v0[:2048].double() - r0).abs() / r0.abs()).max()),
                      max_rel_err_bf16x3_vs_f64=float(((v1[:2048].double() - torch.gather(ref, 1, i1[:2048])).abs() / torch.gather(ref, 1, i1[:2048]).abs()).max()))))
res = {}
for rnd in range(3):
    for prec in ('fp32', 'bf16x3'):
        res.setdefault(prec, []).append(round(time_ms(lambda: score_topk(U, I, 20, precision=prec, **kw), reps=3, warm=1), 2))
print(json.dumps(dict(ms=res)))

"""Developer A/B of two builds of the library in ONE process (same box, alternating):
   python scripts/dev_topk_ab_libs.py libA.so libB.so   -> ms of the full Amazon-like evaluation, k = 20."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import igcn_cf_amd._lib as _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import _csr_to_device, _merge_sorted_csr
from scripts.dev_spmm_bench import time_ms

libs = {os.path.basename(p): C.CDLL(os.path.abspath(p)) for p in sys.argv[1:]}
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
g = torch.Generator(device='cuda').manual_seed(0)
U = torch.randn(ds.n_users, 64, device='cuda', generator=g) * 0.1
I = torch.randn(ds.n_items, 64, device='cuda', generator=g) * 0.1
excl = _merge_sorted_csr(ds.csr('train'), ds.csr('val'))
rp, cl = _csr_to_device(excl[0], excl[1], 'cuda')
users = torch.arange(ds.n_users, device='cuda')
res, outs = {}, {}
for rnd in range(3):
    for name, handle in libs.items():
        _lib._handle, _lib._bound = handle, {}
        for masks in (False, True):
            kw = dict(excl_rowptr=rp, excl_col=cl) if masks else {}
            ms = time_ms(lambda: score_topk(U, I, 20, user_ids=users, precision=os.environ.get('IGCN_AB_PRECISION', 'fp32'), **kw), reps=3, warm=1)
            res.setdefault('%s masks=%d' % (name, masks), []).append(round(ms, 2))
            outs[(name, masks)] = score_topk(U, I, 20, user_ids=users, precision=os.environ.get('IGCN_AB_PRECISION', 'fp32'), **kw)[0]
names = list(libs)
same = all(torch.equal(outs[(names[0], m)], outs[(n, m)]) for n in names[1:] for m in (False, True))
print(json.dumps(dict(ms=res, identical_ids=same)))

"""A/B in one process (IGCN_TOPK_STAGGER read per call): wave priority / start-skew modes of the top-k kernel."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import _csr_to_device, _merge_sorted_csr
from scripts.dev_spmm_bench import time_ms

ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
g = torch.Generator(device='cuda').manual_seed(0)
U = torch.randn(ds.n_users, 64, device='cuda', generator=g) * 0.1
I = torch.randn(ds.n_items, 64, device='cuda', generator=g) * 0.1
excl = _merge_sorted_csr(ds.csr('train'), ds.csr('val'))
rp, cl = _csr_to_device(excl[0], excl[1], 'cuda')
users = torch.arange(ds.n_users, device='cuda')
res = {}
for rnd in range(3):
    for mode in ('0', '1', '2', '3'):
        os.environ['IGCN_TOPK_STAGGER'] = mode
        res.setdefault(mode, []).append(time_ms(lambda: score_topk(U, I, 20, user_ids=users, excl_rowptr=rp, excl_col=cl), reps=3, warm=1))
print(json.dumps({('mode' + m): [round(x, 2) for x in v] for m, v in res.items()}))

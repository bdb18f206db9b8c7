"""Developer: one k=20 timing of the fused top-k on the Amazon-like shape (env knobs vary per run)."""
import json, os, sys
import torch
sys.path.insert(0, '.')
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
for k in (20,):
    ms = min(time_ms(lambda: score_topk(U, I, k, user_ids=users, excl_rowptr=rp, excl_col=cl), reps=5, warm=1) for _ in range(3))
    print(json.dumps(dict(env={k2: v for k2, v in os.environ.items() if k2.startswith('IGCN_')}, k=k, ms=round(ms, 2),
                          users_per_s=round(ds.n_users / ms * 1e3))), flush=True)

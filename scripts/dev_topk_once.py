"""Developer: a few full Amazon-like evaluations (k=20, train+val exclusion), for rocprofv3 passes.
TOPK_MODE=exact (default: the fp32 sweep) | fast (the two-stage path); D=64 (default) | 128."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import _csr_to_device, _merge_sorted_csr

ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
g = torch.Generator(device='cuda').manual_seed(0)
D = int(os.environ.get('D', 64))
U = torch.randn(ds.n_users, D, device='cuda', generator=g) * 0.1
I = torch.randn(ds.n_items, D, device='cuda', generator=g) * 0.1
excl = _merge_sorted_csr(ds.csr('train'), ds.csr('val'))
rp, cl = _csr_to_device(excl[0], excl[1], 'cuda')
users = torch.arange(ds.n_users, device='cuda')
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    score_topk(U, I, 20, user_ids=users, excl_rowptr=rp, excl_col=cl, mode=os.environ.get('TOPK_MODE', 'exact'))
torch.cuda.synchronize()

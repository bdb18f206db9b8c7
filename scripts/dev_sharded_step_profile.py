"""Developer: steps of the row-sharded LightGCN (world size 1, no process group: the exchanges are self-copies) for a
kernel trace:  rocprofv3 --kernel-trace ... -- python3 scripts/dev_sharded_step_profile.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.dist import ShardedLightGCN
from igcn_cf_amd.trainer import DeviceSampler

dev = torch.device('cuda', 0)
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'device': dev})
model = ShardedLightGCN(ds, 64, 3, 0, 1, dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
batches = [b for _, b in zip(range(20), DeviceSampler(ds, dev, 2021).epoch_batches(2048))]


def step(b):
    users, pos, neg = b.t().contiguous().unbind(0)
    terms = model.bpr_loss_terms(users, pos, neg)
    loss = terms[0] + 1e-5 * terms[1]
    opt.zero_grad(); loss.backward(); opt.step()


for b in batches[:5]:
    step(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in batches[5:]:
    step(b)
torch.cuda.synchronize()
print('ms per step', (time.perf_counter() - t0) / 15 * 1e3)

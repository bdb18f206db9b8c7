"""igcn_mark_rows before / after its long rows were split over 8 waves (round 6): the ids of a BPR batch on the Amazon-like graph
(users | positives — popular items — | negatives), same process, both libraries loaded side by side, HIP events, interleaved."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import normalized_adjacency_host

dev = torch.device('cuda', 0)
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021, 'device': dev})
nu, n = ds.n_users, ds.n_users + ds.n_items
rowptr, col, _ = normalized_adjacency_host(ds.train_array, nu, ds.n_items)
rp, cl = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
ta = np.asarray(ds.train_array)
rng = np.random.default_rng(0)
pick = rng.integers(0, len(ta), 2048)
ids = torch.from_numpy(np.concatenate([ta[pick, 0], nu + ta[pick, 1], nu + rng.integers(0, ds.n_items, 2048)]).astype(np.int64)).to(dev)
vp = C.c_void_p
libs = {}
for name, path in (('before', 'igcn_cf_amd/libigcn_hip_before.so'), ('after', 'igcn_cf_amd/libigcn_hip.so')):
    L = C.CDLL(os.path.join(ROOT, path))
    L.igcn_mark_rows.restype = C.c_int
    L.igcn_mark_rows.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_int64, vp]
    libs[name] = L
masks = {k: torch.zeros((2, n), dtype=torch.uint8, device=dev) for k in libs}
st = torch.cuda.current_stream().cuda_stream


def run(name):
    m = masks[name]
    rc = libs[name].igcn_mark_rows(ids.data_ptr(), ids.numel(), rp.data_ptr(), cl.data_ptr(), m[0].data_ptr(), m[1].data_ptr(), n, st)
    assert rc == 0


res = {k: [] for k in libs}
for rnd in range(5):
    for name in libs:
        for _ in range(20):
            run(name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            run(name)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 200 * 1e3)
assert torch.equal(masks['before'], masks['after'])
deg = np.diff(rowptr)[ids.cpu().numpy()]
print(json.dumps({'ids': int(ids.numel()), 'max_row': int(deg.max()), 'mean_row': float(deg.mean()), 'marked_rows': int(masks['after'][1].sum()),
                  'us_before': sorted(res['before'])[2], 'us_after': sorted(res['after'])[2], 'masks_equal': True}))

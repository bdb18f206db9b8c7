"""Developer: the two-stage top-k path against the fp32 sweep — same ids / values, how many users fall back, time."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd import ops
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import _csr_to_device, _merge_sorted_csr
from scripts.dev_spmm_bench import time_ms

g = torch.Generator(device='cuda').manual_seed(0)
D = int(os.environ.get('D', 64))                                         # 64 or 128
from igcn_cf_amd import _lib
if os.environ.get('FAST_MODE'):
    _lib.set_tuning('topk_fast_mode', int(os.environ['FAST_MODE']))      # 1: two bf16 planes, 2 (default): one fp16 item plane
if os.environ.get('EXTRA'):
    _lib.set_tuning('topk_fast_extra', int(os.environ['EXTRA']))
if os.environ.get('WIDE'):
    _lib.set_tuning('topk_fast_wide', int(os.environ['WIDE']))           # d = 128: 1 = two groups per wave, one wave per SIMD
# 1. small exact-arithmetic case (integers: massive ties -> every user must fall back and still be right)
U = torch.randint(-3, 4, (300, D), device='cuda', generator=g).float()
I = torch.randint(-3, 4, (5000, D), device='cuda', generator=g).float()
a = score_topk(U, I, 20, mode='exact'); b = score_topk(U, I, 20, mode='fast')
print(json.dumps({'case': 'integers 300x5000', 'ids_equal': bool(torch.equal(a[0], b[0])), 'vals_equal': bool(torch.equal(a[1], b[1])),
                  'flagged': score_topk.last_flagged}))
# 2. gaussian, several sizes, with masks
for nu, ni, k in ((1000, 20000, 20), (4096, 50000, 5), (513, 3333, 50)):
    U = torch.randn(nu, D, device='cuda', generator=g) * 0.1
    I = torch.randn(ni, D, device='cuda', generator=g) * 0.1
    rng = np.random.default_rng(nu)
    ex = [np.sort(rng.choice(ni, size=int(rng.integers(0, 40)), replace=False)) for _ in range(nu)]
    rowptr = np.zeros(nu + 1, dtype=np.int64); np.cumsum([len(e) for e in ex], out=rowptr[1:])
    col = np.concatenate(ex).astype(np.int32)
    rp, cl = torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda()
    ban = torch.zeros(ni, dtype=torch.uint8, device='cuda'); ban[::7] = 1
    a = score_topk(U, I, k, excl_rowptr=rp, excl_col=cl, banned=ban, mode='exact')
    b = score_topk(U, I, k, excl_rowptr=rp, excl_col=cl, banned=ban, mode='fast')
    print(json.dumps({'case': 'gauss %dx%d k=%d' % (nu, ni, k), 'ids_equal': bool(torch.equal(a[0], b[0])),
                      'vals_equal': bool(torch.equal(a[1], b[1])), 'max_val_diff': float((a[1] - b[1]).abs().max()),
                      'id_mismatch_rows': int((a[0] != b[0]).any(1).sum()), 'flagged': score_topk.last_flagged}))
# 3. the full Amazon-like evaluation
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
U = torch.randn(ds.n_users, D, device='cuda', generator=g) * 0.1
I = torch.randn(ds.n_items, D, device='cuda', generator=g) * 0.1
excl = _merge_sorted_csr(ds.csr('train'), ds.csr('val'))
rp, cl = _csr_to_device(excl[0], excl[1], 'cuda')
for masks in (False, True):
    kw = dict(excl_rowptr=rp, excl_col=cl) if masks else {}
    a = score_topk(U, I, 20, mode='exact', **kw); b = score_topk(U, I, 20, mode='fast', **kw)
    rec = {'case': 'amazon-like full evaluation, masks=%s' % masks, 'ids_equal': bool(torch.equal(a[0], b[0])),
           'vals_equal': bool(torch.equal(a[1], b[1])), 'max_val_diff': float((a[1] - b[1]).abs().max()),
           'id_mismatch_rows': int((a[0] != b[0]).any(1).sum()), 'flagged': score_topk.last_flagged}
    rec['exact_ms'] = round(time_ms(lambda: score_topk(U, I, 20, mode='exact', **kw), reps=5), 3)
    rec['fast_ms'] = round(time_ms(lambda: score_topk(U, I, 20, mode='fast', **kw), reps=5), 3)
    print(json.dumps(rec), flush=True)

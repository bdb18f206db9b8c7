"""Same-process A/B of SpMM dealing schemes on the Amazon-like graph (d = 64, one launch of igcn_spmm_csr_f32):
the round-2 order (rows dealt round-robin over all XCDs) against the XCD plan (graph.xcd_plan_host: per-XCD lists,
rows above a threshold cut at operand-slice boundaries, short rows to the list that holds most of their columns).

    python scripts/dev_xcd_plan_ab.py            # times (HIP events, median of 5 rounds x 50 launches) + checks
    PMC=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d ... -- python3 scripts/dev_xcd_plan_ab.py
                                                 # 3 launches per variant in the printed order, nothing else
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm
from scripts.dev_spmm_bench import time_ms

PMC = os.environ.get('PMC') == '1'
PRESET = os.environ.get('PRESET', 'amazon')
D = int(os.environ.get('DIM', '64'))
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': PRESET, 'seed': 2021})
nu, ni = ds.n_users, ds.n_items
n = nu + ni
rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ni)
blocks = [0, nu, n]

variants = {'legacy': CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=blocks)}
for T in [int(t) for t in os.environ.get('THRESHOLDS', '32,64,96,128,256').split(',')]:
    for assign in ('affinity', 'spread'):
        if assign == 'spread' and T not in (64, 256):
            continue
        variants['xcd_T%d_%s' % (T, assign)] = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=blocks,
                                                         xcd_plan={'threshold': T, 'assign': assign})
if os.environ.get('EXTRA') == '1':
    for tu, ti in ((64, 112), (256, 112), (10 ** 9, 112), (112, 96), (112, 160), (256, 128)):
        variants['xcd_Tuser%d_Titem%d' % (min(tu, 99999), ti)] = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=blocks,
                                                                       xcd_plan={'threshold': [tu, ti], 'segment_len': 112})
    for lo in ('rows_first', 'interleaved'):
        variants['xcd_T112_' + lo] = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=blocks,
                                               xcd_plan={'threshold': 112, 'list_order': lo})
    for sl in (56, 224):
        variants['xcd_T112_seglen%d' % sl] = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=blocks,
                                                       xcd_plan={'threshold': 112, 'segment_len': sl})
    for rc in (0, 2, 8):
        variants['xcd_T112_rowcost%d' % rc] = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=blocks,
                                                        xcd_plan={'threshold': 112, 'row_cost': rc})
# upper bound of any locality scheme: the same structure gathering from an 8192-row (2 MB) table: every gather hits L2
variants['legacy_all_gathers_hit_L2'] = CsrMatrix(rowptr, (col % 8192).astype(np.int32), val, (n, n), 'cuda', order_blocks=blocks)
# one phase at a time (the other block's rows emptied): where the time and the misses are
for name, (lo, hi) in (('legacy_user_rows_only', (0, nu)), ('legacy_item_rows_only', (nu, n))):
    lens = np.diff(rowptr).copy()
    keep = np.zeros(n, dtype=bool); keep[lo:hi] = True
    lens[~keep] = 0
    rp = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=rp[1:])
    e = np.repeat(keep, np.diff(rowptr))
    variants[name] = CsrMatrix(rp, col[e], val[e], (n, n), 'cuda', order_blocks=blocks)

g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(n, D, device='cuda', generator=g) * 0.1
y = torch.empty_like(x)

if PMC:
    names = list(variants)
    for name in names:
        for _ in range(3):
            spmm(variants[name], x, out=y)
    torch.cuda.synchronize()
    print(json.dumps({'pmc_order': names, 'launches_per_variant': 3}))
    sys.exit(0)

ref = spmm(variants['legacy'], x).double()
out = {'preset': PRESET, 'd': D, 'nnz': int(rowptr[-1])}
checks = {}
for name, csr in variants.items():
    if name.startswith('legacy_'):
        continue
    r = spmm(csr, x).double()
    checks[name] = float((r - ref).abs().max() / ref.abs().max())
    assert checks[name] < 1e-5, (name, checks[name])
    r2 = spmm(csr, x).double()
    assert torch.equal(r, r2), name                                 # deterministic
out['max_rel_diff_vs_legacy'] = checks
times = {k: [] for k in variants}
for rnd in range(5):
    for name, csr in variants.items():
        times[name].append(time_ms(lambda: spmm(csr, x, out=y), reps=50))
out['ms'] = {k: round(sorted(v)[2], 4) for k, v in times.items()}
out['plan'] = {k: {'n_long': c.n_long, 'n_segments': c.n_segments,
                   'list_sizes': np.diff(c.xcd_off.cpu().numpy()).tolist() if c.xcd_off is not None else None}
               for k, c in variants.items()}
print(json.dumps(out))

"""Probe for a SLICE-SYNCHRONOUS sweep (round 6 idea, not a product path): would the gathers of an A_hat launch hit L2 if every wave
gathered from the same 1/S of the operand at the same time?

The XCD plan partitions in SPACE (slice s of the operand <-> XCD s) and only pays for rows above ~100 nonzeros: a short row (~20
nonzeros over 8 slices) cannot be cut — 8 partial rows cost more than its misses.  Partitioning in TIME needs no partial sums in
memory (a wave would keep its rows' accumulators and walk the slices 0 .. S-1), but only helps if the waves stay in step.  This probe
measures the ideal of that scheme with kernel boundaries as the barriers: S sub-matrices M_s = the columns of slice s (per phase: user
rows x item slice s, item rows x user slice s; equal gather counts), launched one after the other with the production kernel.
Under rocprofv3 --pmc (scripts/dev_r06_slice_sweep_probe.sh) the sum of FETCH_SIZE over the S launches against the one whole launch
says what the gathers would cost beyond L2; the HIP-event times (each sub-launch visits every row for ~20/S nonzeros: an upper bound
on the overhead of the scheme, not its speed).  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import ops
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host

N_LAUNCH = 4
dev = torch.device('cuda', 0)
d = 64
preset = sys.argv[1] if len(sys.argv) > 1 else 'amazon'
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021, 'device': dev})
nu, n = ds.n_users, ds.n_users + ds.n_items
rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ds.n_items)
nnz = int(rowptr[-1])
g = torch.Generator(device='cpu').manual_seed(2021)
x = (torch.randn(n, d, generator=g) * 0.1).to(dev)
row_of = np.repeat(np.arange(n), np.diff(rowptr))


def slices(S):
    """slice id of every nonzero: per phase the column range cut into S runs of equal gather counts"""
    sl = np.zeros(nnz, dtype=np.int64)
    for lo, hi in ((0, nu), (nu, n)):
        e0, e1 = rowptr[lo], rowptr[hi]
        c = col[e0:e1].astype(np.int64)
        cnt = np.bincount(c - c.min())
        cum = np.cumsum(cnt)
        bounds = c.min() + 1 + np.searchsorted(cum, cum[-1] * np.arange(1, S) / S)
        sl[e0:e1] = np.searchsorted(bounds, c, side='right')
    return sl


variants = []
whole = CsrMatrix(rowptr, col, val, (n, n), dev, order_blocks=[0, nu, n], xcd_plan=XCD_PLAN)
y = torch.empty_like(x)
variants.append(('whole_xcd_plan', [whole]))
plain = CsrMatrix(rowptr, col, val, (n, n), dev, order_blocks=[0, nu, n])
variants.append(('whole_plain_plan', [plain]))
for S in (4, 8, 16):
    sl = slices(S)
    subs = []
    for s in range(S):
        keep = sl == s
        rp = np.concatenate([[0], np.cumsum(np.bincount(row_of[keep], minlength=n))]).astype(np.int64)
        subs.append(CsrMatrix(rp, col[keep], val[keep], (n, n), dev, order_blocks=[0, nu, n]))
    variants.append(('time_slices_%d' % S, subs))
out = {'preset': preset, 'd': d, 'nnz': nnz, 'rows': n, 'n_launch': N_LAUNCH, 'order': [(name, len(ms)) for name, ms in variants], 'variants': {}}
for name, ms in variants:
    def run():
        for m in ms:
            ops.spmm(m, x, out=y)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(N_LAUNCH + 1)]
    e[0].record()
    for j in range(N_LAUNCH):
        run()
        e[j + 1].record()
    torch.cuda.synchronize()
    out['variants'][name] = {'launches': len(ms), 'ms': [e[j].elapsed_time(e[j + 1]) for j in range(N_LAUNCH)],
                             'segments': sum(m.n_segments for m in ms)}
print(json.dumps(out))

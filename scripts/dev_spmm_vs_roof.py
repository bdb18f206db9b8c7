"""Developer: where does the row-per-(sub)wave SpMM lose against the rowless gather roof?  Same nnz and column
popularity, different row structure: the Amazon-like graph, the same with every row cut to the mean degree (uniform
rows, no long rows), rows sorted by length; long-row thresholds."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm

dev = torch.device('cuda')
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
n = ds.n_users + ds.n_items
rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
nnz = int(rowptr[-1])
x = torch.randn(n, 64, device=dev) * 0.1
y = torch.empty_like(x)


def run(tag, rp, c, v, **kw):
    csr = CsrMatrix(rp, c, v, (len(rp) - 1, n), dev, **kw)
    yy = torch.empty((len(rp) - 1, 64), device=dev)
    ms = min(bench.time_ms(lambda: spmm(csr, x, out=yy), 50, 5) for _ in range(3))
    g = bench.gather_roof(dev, csr.col, csr.val, x, len(rp) - 1, 64)
    print(json.dumps(dict(case=tag, rows=len(rp) - 1, nnz=int(rp[-1]), n_long=csr.n_long, n_segments=csr.n_segments,
                          spmm_us=round(ms * 1e3, 1), roof_same_us=round(g['best_same_stream_ms'] * 1e3, 1),
                          frac=round(g['best_same_stream_ms'] / ms, 3))), flush=True)


run('amazon-like, storage order', rowptr, col, val)
ob = [0, ds.n_users, n]
run('amazon-like, ordered', rowptr, col, val, order_blocks=ob)
for lt, sl in ((128, 128), (256, 128), (512, 256), (512, 512), (1024, 256), (1024, 512), (4096, 256)):
    run('ordered, long_threshold=%d segment=%d' % (lt, sl), rowptr, col, val, order_blocks=ob, long_threshold=lt, segment_len=sl)
from igcn_cf_amd import _lib
csr = CsrMatrix(rowptr, col, val, (n, n), dev, order_blocks=ob)
for bpc in (None, 14, 28, 56, 112, 224):
    _lib.set_tuning('spmm_blocks_per_cu', bpc)
    ms = min(bench.time_ms(lambda: spmm(csr, x, out=y), 50, 5) for _ in range(3))
    print(json.dumps(dict(case='ordered, blocks_per_cu=%s' % bpc, spmm_us=round(ms * 1e3, 1))), flush=True)
_lib.set_tuning('spmm_blocks_per_cu', None)

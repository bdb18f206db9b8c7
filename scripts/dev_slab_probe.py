"""Developer probe: column-slab operand layout (every XCD gathers from its own 1/8 of the columns) against the production
SpMM and the rowless gather roof, on the Amazon-like index stream.  See scripts/probes/slab_gather_probe.hip."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import normalized_adjacency_device
from igcn_cf_amd.ops import spmm

lib = ctypes.CDLL(os.path.join(ROOT, 'scripts', 'probes', 'libslab_probe.so'))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
lib.slab_gather.argtypes = [vp, vp, i64, vp, i64, i32, i32, vp, i64, i64, vp]
lib.slab_gather.restype = i32
dev = torch.device('cuda')
preset = sys.argv[1] if len(sys.argv) > 1 else 'amazon'
d = 64
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset})
n = ds.n_users + ds.n_items
csr = normalized_adjacency_device(ds.train_array, ds.n_users, ds.n_items, dev)
nnz = int(csr.col.shape[0])
x = torch.randn(n, d, device=dev) * 0.1
y = torch.empty_like(x)
ms = min(bench.time_ms(lambda: spmm(csr, x, out=y), 50, 5) for _ in range(3))
print(json.dumps(dict(case='production spmm', us=round(ms * 1e3, 1))), flush=True)
g = bench.gather_roof(dev, csr.col, csr.val, x, n, d)
print(json.dumps(dict(case='rowless roof, row layout', us=round(g['best_same_stream_ms'] * 1e3, 1))), flush=True)
st = torch.cuda.current_stream().cuda_stream
# checking mode: per 64-entry chunk of the index stream the weighted sum of the gathered rows, slab-major operand
n_chunks = (nnz + 63) // 64
want = torch.zeros(n_chunks, d, device=dev, dtype=torch.float64)
want.index_add_(0, torch.arange(nnz, device=dev) // 64, x[csr.col.long()].double() * csr.val.double()[:, None])
for piece in (8, 16, 32, 64):
    S = d // piece
    xs = x.view(n, S, piece).permute(1, 0, 2).contiguous()
    ys = torch.zeros(S, n_chunks, piece, device=dev)
    assert lib.slab_gather(csr.col.data_ptr(), csr.val.data_ptr(), nnz, xs.data_ptr(), n, d, piece, ys.data_ptr(), -1, 2048, st) == 0
    torch.cuda.synchronize()
    got = ys.permute(1, 0, 2).reshape(n_chunks, d).double()
    print(json.dumps(dict(case='check', piece_floats=piece, max_abs_err=float((got - want).abs().max()), scale=float(want.abs().max()))), flush=True)
only = [int(a) for a in sys.argv[2:]]
for piece in (only or (8, 16, 32, 64, 32, 16, 8, 64)):
    for bpc in ((16,) if only else (8, 16, 32, 64)):
        blocks = 256 * bpc
        f = lambda: lib.slab_gather(csr.col.data_ptr(), csr.val.data_ptr(), nnz, x.data_ptr(), n, d, piece, y.data_ptr(), n, blocks, st)
        assert f() == 0
        torch.cuda.synchronize()
        ms = min(bench.time_ms(f, 50, 5) for _ in range(3))
        print(json.dumps(dict(case='slab probe', piece_floats=piece, slabs=d // piece, blocks=blocks, us=round(ms * 1e3, 1))), flush=True)

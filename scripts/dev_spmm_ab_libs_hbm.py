import ctypes as C, json, os, sys, torch
sys.path.insert(0, '/root/repo')
import igcn_cf_amd._lib as _lib
from igcn_cf_amd.graph import CsrMatrix
from igcn_cf_amd.ops import spmm
from scripts.dev_spmm_bench import time_ms
from scripts.roofline_s import make_shard
libs = {os.path.basename(p): C.CDLL(os.path.abspath(p)) for p in sys.argv[1:]}
_lib._handle, _lib._bound = next(iter(libs.values())), {}
for d, n_cols, n_rows, nnz in ((128, 12_000_000, 1_500_000, 125_000_000), (64, 12_000_000, 1_500_000, 125_000_000), (128, 206151, 206151, 4360286)):
    rowptr, col, val, total = make_shard(n_rows, n_cols, nnz, 1)
    csr = CsrMatrix(rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy(), (n_rows, n_cols), 'cuda')
    x = torch.randn(n_cols, d, device='cuda') * 0.1
    y = torch.empty(n_rows, d, device='cuda')
    res = {}
    for rnd in range(2):
        for name, h in libs.items():
            _lib._handle, _lib._bound = h, {}
            res.setdefault(name, []).append(round(min(time_ms(lambda: spmm(csr, x, out=y), reps=5, warm=2) for _ in range(2)), 3))
    print(json.dumps(dict(d=d, n_cols=n_cols, nnz=total, ms=res)), flush=True)
    del csr, x, y; torch.cuda.empty_cache()

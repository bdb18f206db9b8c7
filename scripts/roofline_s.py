"""HBM-roofline run of the SpMM at the scale of BASELINE config 5 (10 M users x 2 M items x
500 M edges, d = 128): ONE GPU's share of the row-sharded job — 1/8 of the rows of A_hat
(125 M nonzeros) against the full replicated operand X (12 M x 128 fp32 = 6.1 GB, far beyond
the 256 MiB Infinity Cache), generated on the device.  Also a d = 64 variant and a fraction
sweep.  Prints JSON lines; results are copied into profiles/."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from igcn_cf_amd.graph import CsrMatrix
from igcn_cf_amd.ops import spmm
from scripts.dev_spmm_bench import time_ms


def make_shard(n_rows, n_cols, nnz, seed, zipf=False):
    g = torch.Generator(device='cuda').manual_seed(seed)
    # lognormal row lengths scaled to nnz
    z = torch.randn(n_rows, device='cuda', generator=g)
    w = torch.exp(z)
    deg = torch.clamp((w / w.sum() * nnz).round().long(), min=1)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device='cuda')
    torch.cumsum(deg, 0, out=rowptr[1:])
    total = int(rowptr[-1].item())
    col = torch.randint(0, n_cols, (total,), device='cuda', generator=g, dtype=torch.int32)
    val = torch.rand(total, device='cuda', generator=g) * 0.1
    return rowptr, col, val, total


def main():
    out = []
    shapes = [(128, 12_000_000, 1_500_000, 125_000_000),
              (64, 12_000_000, 1_500_000, 125_000_000),
              (128, 1_500_000, 1_500_000, 125_000_000)]
    if '--full' in sys.argv:        # the whole config-5 matrix on ONE GPU: 12 M x 12 M, 1 G nonzeros, d = 128
        shapes = [(128, 12_000_000, 12_000_000, 1_000_000_000)]
    for d, n_cols, n_rows, nnz in shapes:
        rowptr, col, val, total = make_shard(n_rows, n_cols, nnz, 1)
        csr = CsrMatrix(rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy(), (n_rows, n_cols), 'cuda')
        x = torch.randn(n_cols, d, device='cuda') * 0.1
        y = torch.empty(n_rows, d, device='cuda')
        ms = min(time_ms(lambda: spmm(csr, x, out=y), reps=5, warm=2) for _ in range(2))
        b_alg = total * (8 + 4 * d) + n_rows * (4 * d + 4)
        # spot check against torch on a row sample (fp64)
        rows = torch.randint(0, n_rows, (64,), device='cuda')
        err = 0.0
        for r in rows.tolist():
            s, e = int(rowptr[r]), int(rowptr[r + 1])
            ref = (x[col[s:e].long()].double() * val[s:e].double()[:, None]).sum(0)
            err = max(err, float((y[r].double() - ref).abs().max() / (ref.abs().max() + 1e-30)))
        rec = dict(d=d, n_rows=n_rows, n_cols=n_cols, nnz=total, x_gbytes=n_cols * d * 4 / 1e9, ms=round(ms, 3),
                   gedges_per_s=round(total / ms / 1e6, 2), alg_GBps=round(b_alg / ms / 1e6, 1),
                   frac_of_8TBps=round(b_alg / ms / 1e6 / 8000, 3), n_segments=csr.n_segments, sample_rel_err=err)
        print(json.dumps(rec), flush=True)
        del csr, x, y, rowptr, col, val
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

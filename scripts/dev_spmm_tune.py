"""Developer sweep: long-row threshold / segment length on the Amazon-like graph.
Grid shape and segment order are environment knobs of the library (read once per
process), so this script is run once per setting."""
import json
import os
import sys

import torch

sys.path.insert(0, '.')
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm, propagate_mean
from scripts.dev_spmm_bench import time_ms


def main():
    for preset, zq, d in (('amazon', 150.0, 64), ('amazon', 0.0, 64), ('amazon', 150.0, 128)):
        ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'zipf_q': zq})
        n = ds.n_users + ds.n_items
        rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
        x = torch.randn(n, d, device='cuda') * 0.1
        y = torch.empty_like(x)
        for lt, sl in ((64, 64), (128, 64), (128, 128), (256, 128), (256, 256), (512, 256), (1024, 512)):
            csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', long_threshold=lt, segment_len=sl)
            ms1 = min(time_ms(lambda: spmm(csr, x, out=y), reps=50) for _ in range(3))
            ms3 = min(time_ms(lambda: propagate_mean(csr, x, 3), reps=30) for _ in range(3))
            print(json.dumps(dict(env={k: v for k, v in os.environ.items() if k.startswith('IGCN_')}, zipf_q=zq, d=d,
                                  lt=lt, sl=sl, n_seg=csr.n_segments, ms_layer=round(ms1, 4), ms_3layer=round(ms3, 4),
                                  gedges=round(csr.nnz / ms1 / 1e6, 2))), flush=True)


if __name__ == '__main__':
    main()

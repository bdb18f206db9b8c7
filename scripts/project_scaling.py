"""Single-GPU emulation of ONE rank's per-step work in both multi-GPU modes at N = 1, 2, 4, 8
(weak scaling: graph = N x Amazon-like).  Exchanges are not included: column sharding has none
inside the pass; for row sharding the all-gather volume per rank is printed next to the compute.
Gives the compute side of the scaling curve the driver will measure on the 8-GPU node."""
import json
import sys

import torch

sys.path.insert(0, '.')
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.dist import ShardLayout, local_blocks_host
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import propagate_mean, spmm
from scripts.dev_spmm_bench import time_ms


def main():
    base = SyntheticDataset.PRESETS['amazon']
    d, K = 64, 3
    for world in (1, 2, 4, 8):
        ds = SyntheticDataset({'name': 'SyntheticDataset', 'n_users': base['n_users'] * world, 'n_items': base['n_items'] * world,
                               'n_inter': base['n_inter'] * world, 'seed': 2021})
        n = ds.n_users + ds.n_items
        rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
        nnz = int(rowptr[-1])
        # column sharding: whole graph, d / world columns
        csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda')
        dl = d // world
        x = torch.randn(n, dl, device='cuda') * 0.1
        ms_col = min(time_ms(lambda: propagate_mean(csr, x, K), reps=20) for _ in range(3))
        del csr, x
        # row sharding: rank 0's rows (1/world of users and of items), full operand at d columns
        L = ShardLayout(ds.n_users, ds.n_items, world)
        (urp, ucol, uval), (irp, icol, ival) = local_blocks_host(rowptr, col, val, L, 0)
        cu = CsrMatrix(urp, ucol, uval, (L.bu, L.n_pad), 'cuda')
        ci = CsrMatrix(irp, icol, ival, (L.bi, L.n_pad), 'cuda')
        xf = torch.randn(L.n_pad, d, device='cuda') * 0.1
        yu, yi = torch.empty(L.bu, d, device='cuda'), torch.empty(L.bi, d, device='cuda')
        ms_row = min(time_ms(lambda: (spmm(cu, xf, out=yu), spmm(ci, xf, out=yi)), reps=20) for _ in range(3)) * K
        ag_bytes = (K) * (world - 1) / world * L.n_pad * d * 4            # X_0 + K-1 layer exchanges, received per rank
        print(json.dumps(dict(world=world, nnz=nnz, col_sharded_ms_per_step=round(ms_col, 4),
                              col_sharded_job_gedges=round(K * nnz / ms_col / 1e6, 2),
                              row_sharded_compute_ms_per_step=round(ms_row, 4),
                              row_sharded_allgather_MB_per_rank_per_step=round(ag_bytes / 1e6, 1),
                              row_sharded_job_gedges_if_comm_free=round(K * nnz / ms_row / 1e6, 2))), flush=True)
        del cu, ci, xf, yu, yi
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

"""The launches whose counters profiles/pmc_traffic_config5.json holds (run under rocprofv3 by scripts/profile_config5.sh):
BASELINE config 5's rank-0 share (of 8) — the whole share, its user block alone, its item block alone — each launched
N_LAUNCH times IN THIS ORDER with nothing of the same kernel in between, so that the summariser can tell them apart by
dispatch order.  Prints one JSON line with the sizes and the HIP-event times of the same launches."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from igcn_cf_amd import ops
from igcn_cf_amd.dist import ShardLayout
from igcn_cf_amd.synth import BipartiteGraphDevice

N_LAUNCH = 4                                    # per variant; the summariser drops the first (cold) one
dev = torch.device('cuda', 0)
d, world, rank = 128, 8, 0
g = BipartiteGraphDevice(10_000_000, 2_000_000, 500_000_000, dev, seed=2021)
layout = ShardLayout.balanced(g.rowptr_host(), g.n_users, g.n_items, world)
csr, _ = g.rank_share(layout, rank)
(ulo, uhi) = layout.user_rows(rank)
cu, ci = bench.split_share(csr, uhi - ulo)
x = torch.randn(g.n, d, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 0.1
y = torch.empty(csr.shape[0], d, device=dev)
out = {'d': d, 'world': world, 'rank': rank, 'n_launch': N_LAUNCH, 'order': ['whole', 'user_block', 'item_block'], 'launches': {}}
for name, m, yy in (('whole', csr, y), ('user_block', cu, y[:uhi - ulo]), ('item_block', ci, y[uhi - ulo:])):
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(N_LAUNCH + 1)]
    e[0].record()
    for j in range(N_LAUNCH):
        ops.spmm(m, x, out=yy)
        e[j + 1].record()
    torch.cuda.synchronize()
    ts = [e[j].elapsed_time(e[j + 1]) for j in range(N_LAUNCH)]
    out['launches'][name] = {'rows': m.shape[0], 'nnz': m.nnz, 'n_long_rows': m.n_long, 'ms': ts,
                             'algorithmic_bytes': m.nnz * (8 + 4 * d) + m.shape[0] * (4 * d + 4),
                             # everything once: indices + values, the distinct operand rows touched is not known here -> lower bound
                             'index_and_output_bytes': m.nnz * 8 + m.shape[0] * (4 * d + 8)}
print(json.dumps(out))

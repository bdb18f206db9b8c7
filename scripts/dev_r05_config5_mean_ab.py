"""Developer A/B at BASELINE config 5 (HBM-bound: 10 M x 2 M x 500 M edges, d = 128, whole graph on one GPU, 'halves' exchange):
the K = 3 pass with the factored layer mean (ops.mean_plan) against the reference's association ('stack': three addends in the
last launches), interleaved on one box; and the rank-0-of-8 share's single launches with 0 / 1 / 3 addends."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import ops
from igcn_cf_amd.dist import RowShardedPropagator, ShardLayout
from igcn_cf_amd.synth import BipartiteGraphDevice

dev = torch.device('cuda', 0)
K, d = 3, 128
g = BipartiteGraphDevice(10_000_000, 2_000_000, 500_000_000, dev, seed=2021)
layout = ShardLayout.balanced(g.rowptr_host(), g.n_users, g.n_items, 1)
blocks = g.rank_blocks(layout, 0)
nnz = g.nnz
del g
torch.cuda.empty_cache()
prop = RowShardedPropagator(None, layout.n_users, layout.n_items, K, 0, 1, dev, exchange='halves', layout=layout, local_blocks=blocks, global_nnz=nnz)
gen = torch.Generator(device=dev).manual_seed(100)
eu = torch.randn(layout.n_users, d, device=dev, generator=gen) * 0.1
ei = torch.randn(layout.n_items, d, device=dev, generator=gen) * 0.1
plans = {'factored': ops.mean_plan(K), 'stack': ops.mean_plan(K, 'stack')}


def one_pass(plan):
    prop.load_local_embedding(eu, ei)
    return prop.propagate(plan)


outs = {}
for name, plan in plans.items():
    ru, ri = one_pass(plan)
    outs[name] = (ru.clone(), ri.clone())
diff = max(float((outs['factored'][j] - outs['stack'][j]).abs().max()) for j in (0, 1))
scale = max(float(outs['stack'][j].abs().max()) for j in (0, 1))
res = {k: [] for k in plans}
for rnd in range(4):
    for name, plan in plans.items():
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            one_pass(plan)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t) / 3 * 1e3)
print(json.dumps({'workload': 'config 5 whole graph on one GPU, halves, K=3 d=128', 'nnz': nnz,
                  'pass_ms': {k: [round(v, 2) for v in vs] for k, vs in res.items()},
                  'median_ms': {k: round(sorted(vs)[len(vs) // 2], 2) for k, vs in res.items()},
                  'max_abs_diff_over_max': diff / scale}), flush=True)

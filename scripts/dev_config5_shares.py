"""All eight rank shares of BASELINE config 5 (bipartite 10 M x 2 M x ~500 M edges, d = 128) run one after the other on
one GPU: per-rank rows / nonzeros / milliseconds (the balance ShardLayout.balanced gives) -> one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

out = bench.hbm_bound_leg(torch.device('cuda', 0), reps=3, ranks=tuple(range(8)))
print(json.dumps(out))

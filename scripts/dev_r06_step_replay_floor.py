"""How much of the LightGCN training step (Amazon-like, B = 2048, one captured HIP graph) is the eager work AROUND the graph — the
sampler launch in front of it, the loss clone (and the gap in front of that) behind it?  The trainer's own loop against bare replays of
the same captured graph back to back (same batch every time: the kernels' work is the same to within the batch's rows)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer

dev = torch.device('cuda', 0)
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021, 'device': dev})
torch.manual_seed(2021)
model = get_model({'name': 'LightGCN', 'embedding_size': 64, 'n_layers': 3, 'device': dev}, ds)
trainer = get_trainer({'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': 1e-3, 'l2_reg': 1e-5, 'device': dev, 'n_epochs': 1, 'batch_size': 2048,
                       'dataloader_num_workers': 0, 'test_batch_size': 512, 'topks': [20]}, ds, model)
model.train()
it = trainer.sampler.epoch_node_batches(trainer.batch_size, model.n_users, into=trainer._draw_into(0, lambda b: (3 * b,)))


def loop_ms(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


res = {}
for rnd in range(3):
    res.setdefault('trainer_loop_ms', []).append(loop_ms(lambda: trainer.node_step(next(it)), 100, 20))
    res.setdefault('bare_replays_ms', []).append(loop_ms(trainer._graph.replay, 100, 20))
    res.setdefault('replay_plus_clone_ms', []).append(loop_ms(lambda: (trainer._graph.replay(), trainer._static_loss.clone()), 100, 20))
print(json.dumps({k: sorted(v)[1] for k, v in res.items()}))

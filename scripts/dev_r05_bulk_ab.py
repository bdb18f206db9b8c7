"""Developer (round 5): the candidate sweep's BULK START ("topk_fast_bulk" 1: lists filled from the first two tiles by a radix select
+ one heapify; 0: empty lists) on TRAINED tables — LightGCN on the Amazon-like split after 0 ... E epochs: full evaluation
(propagation + two-stage scoring) with the knob off / on, same process, interleaved; lists against the fp32 sweep."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import get_trainer

dev = torch.device('cuda')
preset = sys.argv[1] if len(sys.argv) > 1 else 'amazon'
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021, 'device': dev})
torch.manual_seed(2021)
model = get_model({'name': 'LightGCN', 'embedding_size': 64, 'n_layers': 3, 'device': dev}, ds)
trainer = get_trainer({'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': 1e-3, 'l2_reg': 1e-5, 'device': dev, 'n_epochs': 1,
                       'batch_size': 2048, 'dataloader_num_workers': 0, 'test_batch_size': 512, 'topks': [20]}, ds, model)


def one(mode):
    model._rep_cache = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = trainer.recommend_all('test', mode=mode)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, rec


for epoch in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    model.eval()
    ts = {0: [], 1: []}
    recs, flagged = {}, {}
    for rnd in range(9):
        for on in (0, 1):
            _lib.set_tuning('topk_fast_bulk', on)
            ms, recs[on] = one('auto')
            flagged[on] = score_topk.last_flagged
            if rnd >= 2:
                ts[on].append(ms)
    _lib.set_tuning('topk_fast_bulk', None)
    _, recx = one('exact')
    med = {on: sorted(ts[on])[len(ts[on]) // 2] for on in (0, 1)}
    print(json.dumps(dict(epochs_trained=epoch, eval_ms_bulk0=round(med[0], 4), eval_ms_bulk1=round(med[1], 4), delta_pct=round(100 * (med[1] / med[0] - 1), 2),
                          min_bulk0=round(min(ts[0]), 4), min_bulk1=round(min(ts[1]), 4), flagged_bulk0=flagged[0], flagged_bulk1=flagged[1],
                          lists_equal_fp32_sweep=bool(torch.equal(recs[0], recx) and torch.equal(recs[1], recx)))), flush=True)
    model.train()
    trainer.train_one_epoch()
    torch.cuda.synchronize()

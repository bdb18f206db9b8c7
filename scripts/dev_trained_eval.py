"""Developer: the evaluation on TRAINED embeddings (the state a recommender is evaluated in: popular items carry long
rows, the candidate sweep's early exit bites) — LightGCN on the Amazon-like split after E epochs: time of recommend_all in
the two-stage modes and the fp32 sweep, users handed to the fp32 sweep."""
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


def timed(mode, planes):
    _lib.set_tuning('topk_fast_mode', planes)
    ts = []
    for i in range(7):
        model._rep_cache = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rec = trainer.recommend_all('test', mode=mode)
        torch.cuda.synchronize()
        if i >= 2:
            ts.append((time.perf_counter() - t0) * 1e3)
    _lib.set_tuning('topk_fast_mode', None)
    ts.sort()
    return round(ts[len(ts) // 2], 3), rec, score_topk.last_flagged


for epoch in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    model.eval()
    ms3, rec3, fl3 = timed('auto', None)
    ms2, rec2, fl2 = timed('auto', 2)
    msx, recx, _ = timed('exact', None)
    with torch.no_grad():
        rep = model.get_rep()
    norms = rep[ds.n_users:].norm(dim=1)
    _, metrics = trainer.eval('test')
    print(json.dumps(dict(epochs_trained=epoch, eval_ms_one_plane=ms3, flagged_one_plane=fl3, eval_ms_two_planes=ms2, flagged_two_planes=fl2,
                          eval_ms_fp32_sweep=msx, lists_equal=bool(torch.equal(rec3, recx) and torch.equal(rec2, recx)),
                          item_norm_max_over_median=round(float(norms.max() / norms.median()), 2), recall20=round(float(metrics['Recall'][20]), 5))), flush=True)
    model.train()
    t0 = time.time()
    trainer.train_one_epoch()
    torch.cuda.synchronize()
    print(json.dumps(dict(epoch_s=round(time.time() - t0, 2))), flush=True)

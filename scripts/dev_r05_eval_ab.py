"""Developer (round 5): full evaluation (propagation + two-stage scoring, trainer.recommend_all) at random init and after E epochs of
LightGCN on the Amazon-like split, for the library IGCN_LIB_PATH names (IGCN_EXPECT_ABI=7: the round-4 build; its missing
igcn_score_topk_fast_finished_max is stood in for here, in the script) — run once per library in one gpurun call."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import _lib
if _lib.EXPECTED_ABI < 8:
    _lib._bound['igcn_score_topk_fast_finished_max'] = lambda batch, with_bound: min(int(batch), 256) if with_bound else 0
for _item in filter(None, os.environ.get('IGCN_TUNE', '').split(',')):        # e.g. IGCN_TUNE=topk_fast_refine=0
    _lib.set_tuning(_item.split('=')[0], int(_item.split('=')[1]))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer

dev = torch.device('cuda')
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021, 'device': dev})
torch.manual_seed(2021)
model = get_model({'name': 'LightGCN', 'embedding_size': 64, 'n_layers': 3, 'device': dev}, ds)
trainer = get_trainer({'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': 1e-3, 'l2_reg': 1e-5, 'device': dev, 'n_epochs': 1,
                       'batch_size': 2048, 'dataloader_num_workers': 0, 'test_batch_size': 512, 'topks': [20]}, ds, model)
res = {'lib': os.environ.get('IGCN_LIB_PATH', 'shipped'), 'tune': os.environ.get('IGCN_TUNE', '')}
for epoch in range(3):
    model.eval()
    ts = []
    for i in range(13):
        model._rep_cache = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer.recommend_all('test', mode='auto')
        torch.cuda.synchronize()
        if i >= 4:
            ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    res['eval_ms_after_%d_epochs' % epoch] = [round(ts[0], 3), round(ts[len(ts) // 2], 3)]
    model.train()
    trainer.train_one_epoch()
    torch.cuda.synchronize()
print(json.dumps(res), flush=True)

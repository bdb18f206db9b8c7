"""Developer: wall time of consecutive full evaluations (recommend_all) on the Amazon-like split, one by one."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer
dev = torch.device('cuda')
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': os.environ.get('PRESET', 'amazon'), 'seed': 2021, 'device': dev})
torch.manual_seed(2021)
model = get_model({'name': 'LightGCN', 'embedding_size': 64, 'n_layers': 3, 'device': dev}, ds)
trainer = get_trainer({'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': 1e-3, 'l2_reg': 1e-5, 'device': dev, 'n_epochs': 1,
                       'batch_size': 2048, 'dataloader_num_workers': 0, 'test_batch_size': 512, 'topks': [20]}, ds, model)
model.eval()
for mode in ('auto', 'exact', 'auto'):
    ts = []
    for _ in range(8):
        model._rep_cache = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer.recommend_all('test', mode=mode)
        torch.cuda.synchronize()
        ts.append(round((time.perf_counter() - t0) * 1e3, 3))
    print(json.dumps({'mode': mode, 'ms': ts}), flush=True)

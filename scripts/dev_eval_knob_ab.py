"""Developer: A/B of one tuning knob on the FULL evaluation as bench.py times it (trainer.recommend_all('test'): propagation + two-stage
scoring, each call timed on its own with a synchronize on both sides, median of 9), alternating the settings in one process.
    python scripts/dev_eval_knob_ab.py topk_fast_warm 0 -1      (-1 = library default)"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer
knob, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
dev = torch.device('cuda')
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021, 'device': dev})
torch.manual_seed(2021)
model = get_model({'name': 'LightGCN', 'embedding_size': 64, 'n_layers': 3, 'device': dev}, ds)
trainer = get_trainer({'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': 1e-3, 'l2_reg': 1e-5, 'device': dev, 'n_epochs': 1,
                       'batch_size': 2048, 'dataloader_num_workers': 0, 'test_batch_size': 512, 'topks': [20]}, ds, model)
if os.environ.get('TRAIN_STEPS'):                      # bench.py evaluates after its training-step measurement (35 steps)
    import bench
    bench.train_step_ms(trainer, int(os.environ['TRAIN_STEPS']), 0)
model.eval()
ts = {v: [] for v in values}
for rnd in range(11):
    for v in values:
        _lib.set_tuning(knob, v if v >= 0 else None)
        model._rep_cache = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer.recommend_all('test')
        torch.cuda.synchronize()
        if rnd >= 2:
            ts[v].append((time.perf_counter() - t0) * 1e3)
_lib.set_tuning(knob, None)
print(json.dumps({'all_ms': {str(v): [round(x, 3) for x in t] for v, t in ts.items()}}))
print(json.dumps({'knob': knob, 'full_eval_ms_median_min': {str(v): [round(sorted(t)[len(t) // 2], 3), round(min(t), 3)] for v, t in ts.items()}}))

"""Developer: LightGCN on the Amazon-like split, 2 epochs, then three scoring calls (two-stage, no masks; MASKS=1: three full
evaluations with the train + val lists masked) for a kernel trace."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import get_trainer
dev = torch.device('cuda')
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021, 'device': dev})
torch.manual_seed(2021)
model = get_model({'name': 'LightGCN', 'embedding_size': 64, 'n_layers': 3, 'device': dev}, ds)
trainer = get_trainer({'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': 1e-3, 'l2_reg': 1e-5, 'device': dev, 'n_epochs': 1,
                       'batch_size': 2048, 'dataloader_num_workers': 0, 'test_batch_size': 512, 'topks': [20]}, ds, model)
for _ in range(2):
    model.train()
    trainer.train_one_epoch()
model.eval()
with torch.no_grad():
    rep = model.get_rep().clone()
U, I = rep[:ds.n_users], rep[ds.n_users:]
users = torch.arange(ds.n_users, device=dev)
torch.cuda.synchronize()
for _ in range(3):
    if os.environ.get('MASKS') == '1':                      # the evaluation as the trainer runs it: propagation + train/val lists masked
        model._rep_cache = None
        trainer.recommend_all('test')
    else:
        score_topk(U, I, 20, user_ids=users, mode='fast')
    torch.cuda.synchronize()

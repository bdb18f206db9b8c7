"""Developer: why does the early exit stop biting after the second epoch?  Scoring alone on the trained tables: with / without
the exclusion lists, exit on / off; how many users have a non-positive k-th best score."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from igcn_cf_amd import _lib
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
rp, cl = trainer._exclusion('test')
users = torch.arange(ds.n_users, device=dev)
for epoch in range(4):
    model.eval()
    with torch.no_grad():
        rep = model.get_rep().clone()
    U, I = rep[:ds.n_users], rep[ds.n_users:]
    rec = {}
    for masks in (False, True):
        kw = dict(excl_rowptr=rp, excl_col=cl) if masks else {}
        for tag, ex, gu in (('exit+give_up', None, None), ('exit', None, 0), ('none', 0, None)):
            _lib.set_tuning('topk_fast_exit', ex)
            _lib.set_tuning('topk_fast_give_up', gu)
            rec['ms_masks%d_%s' % (masks, tag)] = round(bench.time_ms(lambda: score_topk(U, I, 20, user_ids=users, mode='fast', **kw), 5, 2), 3)
            rec['flagged_masks%d_%s' % (masks, tag)] = score_topk.last_flagged
        _lib.set_tuning('topk_fast_exit', None)
        _lib.set_tuning('topk_fast_give_up', None)
        a = score_topk(U, I, 20, user_ids=users, mode='fast', **kw)
        idx, val = score_topk(U, I, 20, user_ids=users, mode='exact', **kw)
        rec['lists_equal_masks%d' % masks] = bool(torch.equal(a[0], idx) and torch.equal(a[1], val))
        rec['users_with_20th_score_le_0_masks%d' % masks] = int((val[:, -1] <= 0).sum())
        rec['min_20th_score_masks%d' % masks] = float(val[:, -1].min())
    rec['user_norm_min_median_max'] = [round(float(x), 5) for x in (U.norm(dim=1).min(), U.norm(dim=1).median(), U.norm(dim=1).max())]
    rec['epochs_trained'] = epoch
    print(json.dumps(rec), flush=True)
    model.train()
    trainer.train_one_epoch()

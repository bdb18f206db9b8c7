"""Developer (round 4): exit checks of the candidate sweep on TRAINED tables — every 24 tiles with give-up from tile 48 (round 3) against
every 6 tiles up to tile 48 with give-up from tile 12.  (An earlier version of this script swept a look-ahead form of the
'far from done' rule, 24 ... 384 tiles: no difference — profiles/r04q_*.)  LightGCN on the
Amazon-like split, epochs 1..3: scoring ms with the train + val lists masked, users handed to the fp32 sweep, lists vs the fp32 sweep."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from igcn_cf_amd import _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import get_trainer
dev = torch.device('cuda')
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': os.environ.get('PRESET', 'amazon'), 'seed': 2021, 'device': dev})
torch.manual_seed(2021)
model = get_model({'name': 'LightGCN', 'embedding_size': 64, 'n_layers': 3, 'device': dev}, ds)
trainer = get_trainer({'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': 1e-3, 'l2_reg': 1e-5, 'device': dev, 'n_epochs': 1,
                       'batch_size': 2048, 'dataloader_num_workers': 0, 'test_batch_size': 512, 'topks': [20]}, ds, model)
rp, cl = trainer._exclusion('test')
for epoch in range(1, 4):
    model.train()
    trainer.train_one_epoch()
    model.eval()
    with torch.no_grad():
        rep = model.get_rep().clone()
    U, I = rep[:ds.n_users], rep[ds.n_users:]
    kw = dict(excl_rowptr=rp, excl_col=cl)
    ref = score_topk(U, I, 20, mode='exact', **kw)
    rec = {'epochs_trained': epoch}
    for la in (0, 3, 6, 12):                                 # 0: round 3's cadence (checks every 24 tiles, give-up from tile 48); n: every n tiles up to 48, give-up from 2 n
        _lib.set_tuning('topk_fast_early_checks', la)
        a = score_topk(U, I, 20, mode='fast', **kw)
        assert torch.equal(a[0], ref[0]) and torch.equal(a[1], ref[1]), la
        ms = sorted(bench.time_ms(lambda: score_topk(U, I, 20, mode='fast', **kw), 5, 1) for _ in range(3))[1]
        rec['early_checks_%d' % la] = {'ms': round(ms, 3), 'handed_over': score_topk.last_flagged}
    _lib.set_tuning('topk_fast_early_checks', None)
    print(json.dumps(rec), flush=True)

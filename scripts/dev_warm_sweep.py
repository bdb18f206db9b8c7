"""Developer (round 4): length of the candidate sweep's warm-up pass ("topk_fast_warm", tiles; 0 = none) — two-stage scoring of the
Amazon-like evaluation (k = 20, train + val lists masked) at random init and on LightGCN tables after 1..2 epochs: ms (median of
three timings of five calls), users handed to the fp32 sweep, lists compared with the fp32 sweep every time."""
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
WARMS = [int(x) for x in os.environ.get('WARMS', '0,32,64,128,256,512').split(',')]
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': os.environ.get('PRESET', 'amazon'), 'seed': 2021, 'device': dev})
torch.manual_seed(2021)
model = get_model({'name': 'LightGCN', 'embedding_size': 64, 'n_layers': 3, 'device': dev}, ds)
trainer = get_trainer({'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': 1e-3, 'l2_reg': 1e-5, 'device': dev, 'n_epochs': 1,
                       'batch_size': 2048, 'dataloader_num_workers': 0, 'test_batch_size': 512, 'topks': [20]}, ds, model)
rp, cl = trainer._exclusion('test')
for epoch in range(0, int(os.environ.get('EPOCHS', 2)) + 1):
    if epoch:
        model.train()
        trainer.train_one_epoch()
    model.eval()
    with torch.no_grad():
        rep = model.get_rep().clone()
    U, I = rep[:ds.n_users], rep[ds.n_users:]
    nrm = I.norm(dim=1).sort(descending=True).values
    print(json.dumps({'epochs_trained': epoch, 'row_norm_at_tile_over_tile_0': {str(t): round(float(nrm[32 * t] / nrm[0]), 3) for t in (6, 12, 32, 64, 128, 256, 1024)}}), flush=True)
    for masks in (True, False):
        kw = dict(excl_rowptr=rp, excl_col=cl) if masks else {}
        ref = score_topk(U, I, 20, mode='exact', **kw)
        rec = {'epochs_trained': epoch, 'masks': masks}
        for w in WARMS:
            _lib.set_tuning('topk_fast_warm', w)
            a = score_topk(U, I, 20, mode='fast', **kw)
            assert torch.equal(a[0], ref[0]) and torch.equal(a[1], ref[1]), w
            ms = sorted(bench.time_ms(lambda: score_topk(U, I, 20, mode='fast', **kw), 5, 1) for _ in range(3))[1]
            rec['warm_%d' % w] = {'ms': round(ms, 3), 'handed_over': score_topk.last_flagged}
        _lib.set_tuning('topk_fast_warm', None)
        print(json.dumps(rec), flush=True)

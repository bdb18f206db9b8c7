"""Developer: full evaluation (recommend_all) before and after training, per model config of the synthetic presets: two-stage
path (default), the same without stragglers giving up, fp32 sweep.  CASES=yelp:2,gowalla:1,amazon:2 (preset:config index)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import _lib, config as cfg
from igcn_cf_amd.dataset import get_dataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import get_trainer

cases = [(c.split(':')[0], int(c.split(':')[1])) for c in os.environ.get('CASES', 'gowalla:1,yelp:2,amazon:2').split(',')]
for preset, index in cases:
    ds_cfg, m_cfg, t_cfg = cfg.get_synthetic_config(torch.device('cuda'), preset)[index]
    ds = get_dataset(ds_cfg)
    torch.manual_seed(2021)
    model = get_model(m_cfg, ds)
    trainer = get_trainer(t_cfg, ds, model)

    def timed(mode, give_up=None):
        _lib.set_tuning('topk_fast_give_up', give_up)
        ts = []
        for i in range(6):
            model._rep_cache = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rec = trainer.recommend_all('test', mode=mode)
            torch.cuda.synchronize()
            if i >= 2:
                ts.append((time.perf_counter() - t0) * 1e3)
        _lib.set_tuning('topk_fast_give_up', None)
        return round(sorted(ts)[len(ts) // 2], 3), rec, score_topk.last_flagged

    for epoch in range(3):
        model.eval()
        ms, rec, fl = timed('auto')
        ms0, rec0, fl0 = timed('auto', 0)
        msx, recx, _ = timed('exact')
        _, metrics = trainer.eval('test')
        print(json.dumps(dict(preset=preset, model=m_cfg['name'], epochs_trained=epoch, eval_ms=ms, handed_over=fl, eval_ms_nobody_gives_up=ms0,
                              eval_ms_fp32_sweep=msx, lists_equal=bool(torch.equal(rec, recx) and torch.equal(rec0, recx)),
                              recall20=round(float(metrics['Recall'][20]), 5))), flush=True)
        model.train()
        trainer.train_one_epoch()
    del model, trainer, ds
    torch.cuda.empty_cache()

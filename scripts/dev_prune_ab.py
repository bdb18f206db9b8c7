"""A/B in one process: LightGCN / IGCN training step with and without propagation pruning."""
import json, sys, time
import torch
sys.path.insert(0, '.')
from igcn_cf_amd import config as cfg
from igcn_cf_amd.dataset import get_dataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer

for preset, index in (('amazon', 1), ('amazon', 2), ('gowalla', 1)):
    ds_cfg, m_cfg, t_cfg = cfg.get_synthetic_config(torch.device('cuda'), preset)[index]
    ds = get_dataset(ds_cfg)
    arms = {}
    for prune in (True, False):
        torch.manual_seed(2021)
        model = get_model(dict(m_cfg, prune_propagation=prune), ds)
        trainer = get_trainer(t_cfg, ds, model)
        model.train()
        its = [b for _, b in zip(range(30), trainer.sampler.epoch_batches(2048))]
        aux = [b for _, b in zip(range(30), trainer.aux_sampler.epoch_batches(2048))] if hasattr(trainer, 'aux_sampler') else None
        arms[prune] = (trainer, its, aux)
    res = {True: [], False: []}
    for rnd in range(6):
        for prune in (True, False):
            trainer, its, aux = arms[prune]
            step = (lambda i: trainer.igcn_step(its[i], aux[i])) if aux else (lambda i: trainer.bpr_step(its[i]))
            for i in range(3):
                step(i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(3, 30):
                step(i)
            torch.cuda.synchronize()
            res[prune].append((time.perf_counter() - t0) / 27 * 1e3)
    print(json.dumps(dict(preset=preset, model=m_cfg['name'], pruned_ms=[round(x, 3) for x in res[True]],
                          full_ms=[round(x, 3) for x in res[False]],
                          median_pruned=round(sorted(res[True])[3], 3), median_full=round(sorted(res[False])[3], 3))), flush=True)

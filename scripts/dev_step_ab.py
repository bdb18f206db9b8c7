"""Developer A/B in one process: LightGCN training step with the SpMM launch knobs toggled (environment, read per call)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import config as cfg
from igcn_cf_amd.dataset import get_dataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer

for preset in ('gowalla', 'amazon'):
    ds_cfg, m_cfg, t_cfg = cfg.get_synthetic_config(torch.device('cuda'), preset)[1]
    ds = get_dataset(ds_cfg)
    torch.manual_seed(2021)
    model = get_model(m_cfg, ds)
    trainer = get_trainer(t_cfg, ds, model)
    model.train()
    batches = [b for _, b in zip(range(80), trainer.sampler.epoch_batches(2048))]
    res = {}
    for rnd in range(3):
        for name, env in (('default', {}), ('one row per wave', {'IGCN_SPMM_MULTIROW': '0'}),
                          ('old: one row per wave, 8 workgroups per CU', {'IGCN_SPMM_MULTIROW': '0', 'IGCN_SPMM_BLOCKS_PER_CU': '8'})):
            for k in ('IGCN_SPMM_MULTIROW', 'IGCN_SPMM_BLOCKS_PER_CU'):
                os.environ.pop(k, None)
            os.environ.update(env)
            for b in batches[:10]:
                trainer.bpr_step(b)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches[10:]:
                trainer.bpr_step(b)
            torch.cuda.synchronize()
            res.setdefault(name, []).append(round((time.perf_counter() - t0) * 1e3 / 70, 4))
    print(json.dumps(dict(preset=preset, train_step_ms=res)), flush=True)

"""Developer (round 5): the captured training steps of BASELINE configs 2 / 3 / 4 and get_rep, for the library IGCN_LIB_PATH names
(default: the shipped one; IGCN_EXPECT_ABI=7 lets the round-4 build load) — run once per library in the same gpurun call."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from igcn_cf_amd import config as cfg
from igcn_cf_amd.dataset import get_dataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer

dev = torch.device('cuda', 0)
res = {'lib': os.environ.get('IGCN_LIB_PATH', 'shipped')}
for preset, index in (('gowalla', 1), ('yelp', 2), ('amazon', 1)):
    ds_cfg, m_cfg, t_cfg = cfg.get_synthetic_config(dev, preset)[index]
    ds = get_dataset(ds_cfg)
    steps, reps = [], []
    for rnd in range(3):
        torch.manual_seed(2021)
        model = get_model(dict(m_cfg, embedding_size=64, n_layers=3), ds)
        trainer = get_trainer(dict(t_cfg, hip_graph=True), ds, model)
        if index == 2:
            model.train()
            it = zip(trainer.sampler.epoch_node_batches(trainer.batch_size, model.n_users, into=trainer._draw_into(0, lambda b: (3 * b,))),
                     trainer.aux_sampler.epoch_batches(trainer.batch_size, into=trainer._draw_into(1, lambda b: (b, 3))))
            for _ in range(10):
                trainer.igcn_node_step(*next(it))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                trainer.igcn_node_step(*next(it))
            torch.cuda.synchronize()
            steps.append(round((time.perf_counter() - t0) * 10, 4))
        else:
            steps.append(round(bench.train_step_ms(trainer, 100, 10), 4))
        model.eval()
        with torch.no_grad():
            def rep():
                model._rep_cache = None
                return model.get_rep()
            reps.append(round(bench.time_ms(rep, 200, 10), 4))
        del trainer, model
    res[preset] = {'train_step_ms': steps, 'get_rep_ms': reps}
print(json.dumps(res), flush=True)

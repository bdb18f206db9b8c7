"""Developer (round 5): does adding the cut rows up inside the launch ("spmm_fold" 1) pay in a CAPTURED training step, where every
kernel is a graph node of ~4.7 us whatever it does?  LightGCN on the Gowalla- and Amazon-like splits, IGCN on the Yelp-like split
(bench.py's own step loops), knob 0 / 1 set before the trainer captures; plus get_rep in eval mode."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from igcn_cf_amd import _lib
from igcn_cf_amd import config as cfg
from igcn_cf_amd.dataset import get_dataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer

dev = torch.device('cuda', 0)
for preset, index in (('gowalla', 1), ('yelp', 2), ('amazon', 1)):
    ds_cfg, m_cfg, t_cfg = cfg.get_synthetic_config(dev, preset)[index]
    ds = get_dataset(ds_cfg)
    res = {'preset': preset, 'model': m_cfg['name']}
    for rnd in range(2):
        for on in (0, 1):
            _lib.set_tuning('spmm_fold', on)
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
                ms = (time.perf_counter() - t0) * 10
            else:
                ms = bench.train_step_ms(trainer, 100, 10)
            res.setdefault('train_step_ms_fold%d' % on, []).append(round(ms, 4))
            model.eval()
            with torch.no_grad():
                def rep():
                    model._rep_cache = None
                    return model.get_rep()
                res.setdefault('get_rep_ms_fold%d' % on, []).append(round(bench.time_ms(rep, 200, 10), 4))
            res['n_segments_A_hat'] = model.norm_adj.n_segments
            del trainer, model
    print(json.dumps(res), flush=True)
_lib.set_tuning('spmm_fold', None)

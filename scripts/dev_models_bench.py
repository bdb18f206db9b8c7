"""Developer: per-model train-step / eval timings on the synthetic presets (configs 1-4 of BASELINE.json)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, '.')
from igcn_cf_amd import config as cfg
from igcn_cf_amd.dataset import get_dataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer


def main():
    cases = (('gowalla', 0), ('gowalla', 1), ('yelp', 2), ('yelp', 6), ('amazon', 1), ('amazon', 2))
    if os.environ.get('CASES'):                      # e.g. CASES=yelp:2,yelp:6
        cases = tuple((c.split(':')[0], int(c.split(':')[1])) for c in os.environ['CASES'].split(','))
    for preset, index in cases:
        ds_cfg, m_cfg, t_cfg = cfg.get_synthetic_config(torch.device('cuda'), preset)[index]
        if os.environ.get('HIP_GRAPH') == '0':           # graph models: launch the step's kernels one by one
            t_cfg = dict(t_cfg, hip_graph=False)
        t0 = time.time()
        ds = get_dataset(ds_cfg)
        t_ds = time.time() - t0
        torch.manual_seed(2021)
        t0 = time.time()
        model = get_model(m_cfg, ds)
        trainer = get_trainer(t_cfg, ds, model)
        torch.cuda.synchronize()
        t_build = time.time() - t0
        model.train()
        graph_model = hasattr(model, 'bpr_loss_nodes')
        its = [b for _, b in zip(range(40), trainer.sampler.epoch_node_batches(2048, ds.n_users) if graph_model
                                 else trainer.sampler.epoch_batches(2048))]
        aux = [b for _, b in zip(range(40), trainer.aux_sampler.epoch_batches(2048))] if hasattr(trainer, 'aux_sampler') else None
        if aux:
            step = lambda i: trainer.igcn_node_step(its[i], aux[i])
        else:
            step = (lambda i: trainer.node_step(its[i])) if graph_model else (lambda i: trainer.bpr_step(its[i]))
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(5, 40):
            step(i)
        torch.cuda.synchronize()
        ms_step = (time.time() - t0) / 35 * 1e3
        model.eval()
        trainer.eval('test')
        torch.cuda.synchronize()
        t0 = time.time()
        model._rep_cache = None
        _, metrics = trainer.eval('test')
        ms_eval = (time.time() - t0) * 1e3
        print(json.dumps(dict(preset=preset, model=m_cfg['name'], hip_graph=bool(t_cfg.get('hip_graph', True)), n_users=ds.n_users, n_items=ds.n_items, train_pairs=len(ds),
                              dataset_s=round(t_ds, 2), build_s=round(t_build, 2), train_step_ms=round(ms_step, 3),
                              steps_per_epoch=-(-len(ds) // 2048), epoch_train_s=round(ms_step * (-(-len(ds) // 2048)) / 1e3, 3),
                              eval_test_ms=round(ms_eval, 1), recall20=float(metrics['Recall'][20]))), flush=True)


if __name__ == '__main__':
    main()

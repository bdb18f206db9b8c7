"""Developer: N training steps of one (preset, index) config, for rocprofv3 --kernel-trace --stats."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import config as cfg
from igcn_cf_amd.dataset import get_dataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.trainer import get_trainer

preset, index = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ('amazon', 2)
ds_cfg, m_cfg, t_cfg = cfg.get_synthetic_config(torch.device('cuda'), preset)[index]
ds = get_dataset(ds_cfg)
torch.manual_seed(2021)
model = get_model(m_cfg, ds)
trainer = get_trainer(t_cfg, ds, model)
model.train()
if not hasattr(model, 'bpr_loss_terms_nodes'):     # MF: triplet batches
    its = trainer.sampler.epoch_batches(2048, into=trainer._draw_into(0, lambda b: (b, 3)))
    for i in range(60):
        trainer.bpr_step(next(its))
else:
    its = trainer.sampler.epoch_node_batches(2048, model.n_users, into=trainer._draw_into(0, lambda b: (3 * b,)))
    aux = trainer.aux_sampler.epoch_batches(2048, into=trainer._draw_into(1, lambda b: (b, 3))) if hasattr(trainer, 'aux_sampler') else None
    for i in range(60):                               # the loop of train_one_epoch
        trainer.igcn_node_step(next(its), next(aux)) if aux else trainer.node_step(next(its))
torch.cuda.synchronize()

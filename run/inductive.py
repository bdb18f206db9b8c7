"""Inductive scenarios of the paper, same steps as the reference's run/dropit/igcn_dropit.py:26-37
and run/dropui/igcn_dropui.py:26-35: train on a reduced dataset, then swap in the full graph /
template-feature matrix on the LIVE model (no retraining) and evaluate.

  dropit: every user's train list cut to its first 80 % -> train -> evaluate with the old and the
          updated interactions;
  dropui: first 80 % of users and items -> train -> add the new users/items (they get
          representations from the trained templates) -> the six masked inductive evaluations.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import config as cfg                      # noqa: E402
from igcn_cf_amd.dataset import dropit_dataset, get_dataset, resize_dataset   # noqa: E402
from igcn_cf_amd.model import get_model                    # noqa: E402
from igcn_cf_amd.trainer import get_trainer                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenario', default='dropui', choices=['dropit', 'dropui'])
    ap.add_argument('--dataset', default='gowalla', choices=['gowalla', 'yelp', 'amazon'])
    ap.add_argument('--index', type=int, default=2, help='2 = IGCN, 6 = IMF')
    ap.add_argument('--path', default=None, help='directory with the FULL dataset (train/val/test.txt); default: synthetic')
    ap.add_argument('--epochs', type=int, default=5)
    ap.add_argument('--ratio', type=float, default=0.8)
    args = ap.parse_args()

    torch.manual_seed(2021)
    device = torch.device('cuda')
    dataset_config, model_config, trainer_config = cfg.get_synthetic_config(device, args.dataset)[args.index]
    if args.path:
        dataset_config = {'name': 'ProcessedDataset', 'path': args.path, 'device': device}
    trainer_config = dict(trainer_config, n_epochs=args.epochs)
    new_dataset = get_dataset(dataset_config)
    dataset = dropit_dataset(new_dataset, args.ratio) if args.scenario == 'dropit' else resize_dataset(new_dataset, args.ratio)

    model = get_model(model_config, dataset)
    trainer = get_trainer(trainer_config, dataset, model)
    trainer.train(verbose=True)

    model.config['dataset'] = new_dataset
    if args.scenario == 'dropit':
        trainer = get_trainer(trainer_config, new_dataset, model)
        results, _ = trainer.eval('test')
        print('Previous interactions test result. {:s}'.format(results))
        model.norm_adj = model.generate_graph(new_dataset)
        model.feat_mat, _, _, model.row_sum = model.generate_feat(new_dataset, is_updating=True)
        model.update_feat_mat()
        results, _ = trainer.eval('test')
        print('Updated interactions test result. {:s}'.format(results))
    else:
        model.n_users, model.n_items = new_dataset.n_users, new_dataset.n_items
        model.norm_adj = model.generate_graph(new_dataset)
        model.feat_mat, _, _, model.row_sum = model.generate_feat(new_dataset, is_updating=True)
        model.update_feat_mat()
        trainer = get_trainer(trainer_config, new_dataset, model)
        print('Inductive results.')
        trainer.inductive_eval(dataset.n_users, dataset.n_items)


if __name__ == '__main__':
    main()

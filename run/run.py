"""Transductive entry point: same five steps as the reference's run/run.py:10-26
(pick a config triple, dataset -> model -> trainer, train, test).  Without the
paper's data files, pass --synthetic to run on a seeded synthetic split."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import config as cfg                      # noqa: E402
from igcn_cf_amd.dataset import get_dataset                # noqa: E402
from igcn_cf_amd.model import get_model                    # noqa: E402
from igcn_cf_amd.trainer import get_trainer                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataset', default='gowalla', choices=['gowalla', 'yelp', 'amazon'])
    ap.add_argument('--index', type=int, default=2, help='position in the config list (0 MF, 1 LightGCN, 2 IGCN, 6 IMF)')
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--path', default=None, help='directory with train/val/test.txt')
    ap.add_argument('--epochs', type=int, default=None)
    ap.add_argument('--seed', type=int, default=2021)
    args = ap.parse_args()

    torch.manual_seed(args.seed)
    device = torch.device('cuda')
    if args.synthetic:
        triples = cfg.get_synthetic_config(device, args.dataset, seed=args.seed)
    else:
        triples = getattr(cfg, 'get_%s_config' % args.dataset)(device)
    dataset_config, model_config, trainer_config = triples[args.index]
    if args.path:
        dataset_config = dict(dataset_config, path=args.path)
    if args.epochs is not None:
        trainer_config = dict(trainer_config, n_epochs=args.epochs)

    dataset = get_dataset(dataset_config)
    model = get_model(model_config, dataset)
    trainer = get_trainer(trainer_config, dataset, model)
    trainer.train(verbose=True)
    results, _ = trainer.eval('test')
    print('Test result. {:s}'.format(results))


if __name__ == '__main__':
    main()

"""Config triples (dataset_config, model_config, trainer_config) with the
reference's dict schema and list positions (config.py:1-207): index 0 = MF,
1 = LightGCN, 2 = IGCN, 6 = IMF.  Positions that hold baselines outside the
hot path (3 ItemKNN, 4 NGCF, 5 MultiVAE, 7 IMCGAE, 8 IDCF_LGCN, 9 NeuMF) are
None here.  Hyper-parameters are the reference's published settings."""

_COMMON = dict(optimizer='Adam', n_epochs=1000, batch_size=2048, dataloader_num_workers=6,
               test_batch_size=512, topks=[20])

# dataset -> {model: (model overrides, trainer overrides)}
_TABLE = {
    'Gowalla': {'MF': ({}, dict(lr=1.e-4, l2_reg=1.e-3)),
                'LightGCN': ({}, dict(lr=1.e-3, l2_reg=1.e-4)),
                'IGCN': (dict(dropout=0.3), dict(lr=1.e-3, l2_reg=0., aux_reg=0.01)),
                'IMF': (dict(dropout=0.1), dict(lr=1.e-3, l2_reg=1.e-5, aux_reg=0.1))},
    'Yelp': {'MF': ({}, dict(lr=1.e-3, l2_reg=1.e-3)),
             'LightGCN': ({}, dict(lr=1.e-3, l2_reg=1.e-4)),
             'IGCN': (dict(dropout=0.3), dict(lr=1.e-3, l2_reg=0., aux_reg=0.01)),
             'IMF': (dict(dropout=0.5), dict(lr=1.e-3, l2_reg=1.e-5, aux_reg=0.01))},
    'Amazon': {'MF': ({}, dict(lr=1.e-3, l2_reg=1.e-4)),
               'LightGCN': ({}, dict(lr=1.e-3, l2_reg=1.e-5)),
               'IGCN': (dict(dropout=0.), dict(lr=1.e-3, l2_reg=0., aux_reg=0.01)),
               'IMF': (dict(dropout=0.3), dict(lr=1.e-3, l2_reg=1.e-5, aux_reg=0.1))},
}
_POSITIONS = ['MF', 'LightGCN', 'IGCN', None, None, None, 'IMF', None, None, None]
_TRAINER = {'MF': 'BPRTrainer', 'LightGCN': 'BPRTrainer', 'IGCN': 'IGCNTrainer', 'IMF': 'IGCNTrainer'}
_MODEL_BASE = {'MF': dict(embedding_size=64),
               'LightGCN': dict(embedding_size=64, n_layers=3),
               'IGCN': dict(embedding_size=64, n_layers=3, feature_ratio=1.),
               'IMF': dict(embedding_size=64, n_layers=0, feature_ratio=1.)}


def _build(dataset, device, dataset_config=None):
    if dataset_config is None:
        dataset_config = {'name': 'ProcessedDataset', 'path': 'data/%s/time' % dataset, 'device': device}
    out = []
    for name in _POSITIONS:
        if name is None:
            out.append(None)
            continue
        m_over, t_over = _TABLE[dataset][name]
        model_config = dict(name=name, device=device, **_MODEL_BASE[name])
        model_config.update(m_over)
        trainer_config = dict(name=_TRAINER[name], device=device, **_COMMON)
        trainer_config['topks'] = list(_COMMON['topks'])
        trainer_config.update(t_over)
        out.append((dataset_config, model_config, trainer_config))
    return out


def get_gowalla_config(device):
    return _build('Gowalla', device)


def get_yelp_config(device):
    return _build('Yelp', device)


def get_amazon_config(device):
    return _build('Amazon', device)


def get_synthetic_config(device, preset='amazon', **dataset_kw):
    """Same triples on a seeded synthetic split of the named shape (no data files here)."""
    base = {'gowalla': 'Gowalla', 'yelp': 'Yelp', 'amazon': 'Amazon'}[preset]
    ds = dict(name='SyntheticDataset', preset=preset, device=device)
    ds.update(dataset_kw)
    return _build(base, device, ds)

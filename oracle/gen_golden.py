"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference).  It imports the
reference modules that import unmodified on CPU — ``utils``, ``dataset``,
``trainer`` — and records their outputs on small seeded inputs.  Where a reference
trainer needs a model object, a small hand-written one defined in this file is
passed in; it is an *input* to the reference code and is recorded with the outputs.

``model.py`` (round 3): its one missing import is DGL (``model.py:11``), which the image
lacks.  NO stand-in for DGL exists here: ``model_fixtures`` registers an INERT module
object under the name ``dgl`` — it has no attributes; touching any raises and is logged —
so that the ``import dgl`` statement passes, runs only the functions of ``model.py`` that
never call DGL (graph / feature construction, anneal, dropout of a sparse matrix,
bpr_forward / predict with ``get_rep`` replaced by a recorded tensor, save / load), and
asserts at the end that the placeholder was never touched.  Every recorded number is
therefore computed by the reference's own scipy / torch code.  ``get_rep`` and
``inductive_rep_layer`` (the two gspmm callers) cannot be run and stay definitional.

Outputs are data only (inputs + expected outputs): .npz arrays and toy datasets
in the reference's own text format.  Usage:  python oracle/gen_golden.py
"""
import io
import os
import random
import sys
import contextlib

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def write_lists(path, data):
    with open(path, 'w') as f:
        for u, items in enumerate(data):
            f.write(' '.join([str(u)] + [str(i) for i in items]) + '\n')


def make_toy(name, n_users, n_items, seed, quirks):
    """Seeded toy split in the reference text format (dataset.py:154-164)."""
    rng = np.random.RandomState(seed)
    pop = 1. / np.arange(1, n_items + 1) ** 0.8
    pop /= pop.sum()
    train, val, test = [], [], []
    for u in range(n_users):
        n = int(rng.randint(4, 12))
        items = rng.choice(n_items - 2 if quirks else n_items, size=n, replace=False, p=None if not quirks else None)
        items = [int(i) for i in items]
        n_tr = max(1, int(n * 0.7))
        n_te = max(1, int(n * 0.2))
        train.append(items[:n_tr]); val.append(items[n_tr:n - n_te]); test.append(items[n - n_te:])
    if quirks:
        # users with an empty train list / empty val / empty test list
        train[3] = []; train[17] = []; val[5] = []; test[7] = []; test[3] = []
        # duplicate (user, item) pairs in train -> adjacency value 2
        train[0] = train[0] + [train[0][0]]
        train[9] = train[9] + [train[9][1], train[9][1]]
        # item n_items-2 has train degree exactly 1; item n_items-1 appears only in test
        train[11] = train[11] + [n_items - 2]
        test[12] = test[12] + [n_items - 1]
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    write_lists(os.path.join(d, 'train.txt'), train)
    write_lists(os.path.join(d, 'val.txt'), val)
    write_lists(os.path.join(d, 'test.txt'), test)
    return d


class RecordingModel(torch.nn.Module):
    """Hand-written model object handed to the reference trainers (an input).

    ``predict`` returns rows of a fixed score matrix; ``bpr_forward`` gathers
    rows of a fixed [n_users + n_items, d] parameter and records its arguments.
    """

    def __init__(self, n_users, n_items, d, seed, template=False):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.name = 'RecordingModel'
        self.trainable = True
        self.n_users, self.n_items = n_users, n_items
        self.rep = torch.nn.Parameter(torch.randn(n_users + n_items, d, generator=g) * 0.3)
        # scores are a rank-16 product so that a kernel can reproduce them from the factors
        self.score_u = torch.randn(n_users, 16, generator=g)
        self.score_i = torch.randn(n_items, 16, generator=g)
        self.score_i[4] = self.score_i[2]                 # exact ties (two identical items)
        self.scores = (self.score_u.double() @ self.score_i.double().t()).float()
        self.calls = []
        if template:
            self.embedding = torch.nn.Embedding(n_users + n_items + 2, d)
            with torch.no_grad():
                self.embedding.weight.copy_(torch.randn(n_users + n_items + 2, d, generator=g) * 0.3)
            self.w = torch.nn.Parameter(torch.rand(d, generator=g) + 0.5)
            self.user_map = {u: u for u in range(n_users)}
            self.item_map = {i: i for i in range(n_items)}
            self.anneal_calls = 0

    def predict(self, users):
        return self.scores[users].clone()

    def bpr_forward(self, users, pos_items, neg_items):
        u, p, n = self.rep[users], self.rep[self.n_users + pos_items], self.rep[self.n_users + neg_items]
        l2 = torch.norm(u, p=2, dim=1) ** 2 + torch.norm(p, p=2, dim=1) ** 2 + torch.norm(n, p=2, dim=1) ** 2
        self.calls.append(dict(users=users.numpy().copy(), pos=pos_items.numpy().copy(),
                               neg=neg_items.numpy().copy()))
        return u, p, n, l2

    def feat_mat_anneal(self):
        self.anneal_calls += 1


# scripted training runs: (n_epochs, max_patience, val_interval, validation NDCG@topks[0] per validation)
TRAIN_SCRIPTS = [
    (12, 3, 1, [0.10, 0.12, 0.11, 0.13, 0.13, 0.12, 0.125, 0.11, 0.2, 0.3, 0.1, 0.1]),     # stops early at patience 3
    (7, 4, 2, [0.30, 0.20, 0.25]),                                                          # val every 2nd epoch, runs out
    (5, 50, 1, [0.01, 0.02, 0.03, 0.04, 0.05]),                                             # improves every epoch
]


class ScriptedModel(torch.nn.Module):
    """Input to the reference trainer: records save / load, writes a real (tiny) file so os.remove works."""

    def __init__(self, events, trainable=True):
        super().__init__()
        self.name, self.trainable, self.events = 'Scripted', trainable, events

    def save(self, path):
        self.events.append('save ' + os.path.basename(path))
        open(path, 'w').close()

    def load(self, path):
        self.events.append('load ' + os.path.basename(path))


def run_train_script(trainer_cls, dataset, script, trainable=True, workdir=None):
    """Runs trainer_cls.train() with scripted epochs; returns (events, returned value, epochs run).  Shared by the
    generator (reference trainer) and — re-stated — by tests/test_host_cpu.py (this package's trainer)."""
    n_epochs, patience, interval, ndcgs = script
    events = []
    model = ScriptedModel(events, trainable)

    class Scripted(trainer_cls):
        def train_one_epoch(self_inner):
            events.append('epoch %d' % self_inner.epoch)
            return 1.0 / (1 + self_inner.epoch)

        def eval(self_inner, val_or_test, banned_items=None):
            events.append('eval ' + val_or_test)
            v = ndcgs[min(self_inner._n_val, len(ndcgs) - 1)] if val_or_test == 'val' else 0.5
            if val_or_test == 'val':
                self_inner._n_val += 1
            m = {name: {k: np.float64(v) for k in self_inner.topks} for name in ('Precision', 'Recall', 'NDCG')}
            return 'scripted ', m
    cfg = {'name': 'Scripted', 'dataset': dataset, 'model': model, 'topks': [5, 10], 'device': 'cpu', 'n_epochs': n_epochs,
           'max_patience': patience, 'val_interval': interval, 'test_batch_size': 7}
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            tr = Scripted(cfg)
            tr._n_val = 0
            ret = tr.train(verbose=True)
        left = sorted(os.listdir('checkpoints')) if os.path.isdir('checkpoints') else []
    finally:
        os.chdir(cwd)
    return events, float(ret), left


def train_protocol_fixture(ref_trainer, ds):
    import tempfile
    g = {}
    for j, script in enumerate(TRAIN_SCRIPTS):
        with tempfile.TemporaryDirectory() as tmp:
            events, ret, left = run_train_script(ref_trainer.BasicTrainer, ds, script, True, tmp)
        g['train%d_script' % j] = np.array(list(script[:3]) + list(script[3]), dtype=np.float64)
        g['train%d_events' % j] = np.array(events)
        g['train%d_return' % j] = np.float64(ret)
        g['train%d_checkpoints_left' % j] = np.array(left)
    with tempfile.TemporaryDirectory() as tmp:                        # a non-trainable model is only validated
        events, ret, left = run_train_script(ref_trainer.BasicTrainer, ds, TRAIN_SCRIPTS[0], False, tmp)
    g['train_nontrainable_events'], g['train_nontrainable_return'] = np.array(events), np.float64(ret)
    return g


class _InertPlaceholder(type(sys)):
    """What ``import dgl`` resolves to in model_fixtures(): a module object with NOTHING in it.  Any attribute the
    reference asks of it raises (and is logged), so no number in a fixture can come from here."""
    touched = []

    def __getattr__(self, name):
        if not (name.startswith('__') and name.endswith('__')):        # introspection by importlib / inspect is not use
            _InertPlaceholder.touched.append(name)
        raise AttributeError('inert placeholder: dgl.%s does not exist (DGL is not installed)' % name)


def _coo(t):
    t = t.coalesce()
    return t.indices().numpy().copy(), t.values().detach().numpy().copy()


def _map_arrays(m):
    k = np.array([int(x) for x in m.keys()], dtype=np.int64)
    v = np.array([int(x) for x in m.values()], dtype=np.int64)
    return k, v


def model_fixtures():
    """tests/golden/<toy>_model.npz: outputs of the DGL-free functions of the reference's model.py
    (model.py:52-72, :85-94, :108-123, :263-275, :293-299, :355-421, :448-466)."""
    sys.path.insert(0, REF)
    assert 'dgl' not in sys.modules
    sys.modules['dgl'] = _InertPlaceholder('dgl')
    quiet = contextlib.redirect_stdout(io.StringIO())
    import warnings
    warnings.filterwarnings('ignore')
    # model.py:460 calls torch.load(path) on a checkpoint holding dicts with numpy keys: torch >= 2.6 refuses that by
    # default (weights_only); the documented environment switch restores the behaviour the reference was written for
    os.environ['TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD'] = '1'
    with quiet:
        import dataset as ref_dataset      # noqa: E402  (reference, unmodified)
        import model as ref_model          # noqa: E402  (reference, unmodified)
    d = 8

    def dataset_at(path):
        with quiet:
            return ref_dataset.get_dataset({'name': 'ProcessedDataset', 'path': path, 'device': 'cpu'})

    def build(name, ds, **kw):
        cfg = {'name': name, 'embedding_size': d, 'n_layers': 2, 'device': 'cpu'}
        cfg.update(kw)
        with quiet:
            return ref_model.get_model(cfg, ds)

    for toy in ('toy_a', 'toy_b'):
        g = {}
        ds = dataset_at(os.path.join(OUT, toy))
        n = ds.n_users + ds.n_items
        gen = torch.Generator().manual_seed(31)
        users = torch.randint(0, ds.n_users, (40,), generator=gen)
        pos = torch.randint(0, ds.n_items, (40,), generator=gen)
        neg = torch.randint(0, ds.n_items, (40,), generator=gen)
        users[1] = users[0]; pos[3] = neg[3]                         # duplicate user, positive == negative
        rep = torch.randn(n, d, generator=gen) * 0.4
        pred_users = torch.arange(0, ds.n_users, 3)
        g['users'], g['pos'], g['neg'], g['rep'], g['pred_users'] = (t.numpy() for t in (users, pos, neg, rep, pred_users))

        # ---- MF (model.py:52-72)
        torch.manual_seed(41)
        mf = build('MF', ds)
        g['mf_user_emb'], g['mf_item_emb'] = mf.user_embedding.weight.detach().numpy().copy(), mf.item_embedding.weight.detach().numpy().copy()
        out = mf.bpr_forward(users, pos, neg)
        for tag, t in zip(('u', 'p', 'n', 'l2'), out):
            g['mf_bpr_' + tag] = t.detach().numpy().copy()
        g['mf_predict'] = mf.predict(pred_users).detach().numpy().copy()
        g['mf_state_keys'] = np.array(list(mf.state_dict().keys()))

        # ---- LightGCN (model.py:75-123)
        torch.manual_seed(42)
        lg = build('LightGCN', ds)
        g['lgcn_adj_indices'], g['lgcn_adj_values'] = _coo(lg.norm_adj)
        g['lgcn_adj_shape'] = np.array(lg.norm_adj.shape)
        g['lgcn_emb'] = lg.embedding.weight.detach().numpy().copy()
        lg.get_rep = lambda: rep                                     # get_rep needs DGL: replaced by a recorded tensor (an input)
        out = lg.bpr_forward(users, pos, neg)
        for tag, t in zip(('u', 'p', 'n', 'l2'), out):
            g['lgcn_bpr_' + tag] = t.detach().numpy().copy()
        g['lgcn_predict'] = lg.predict(pred_users).detach().numpy().copy()
        g['lgcn_state_keys'] = np.array(list(lg.state_dict().keys()))

        # ---- IGCN (model.py:355-466) at several template ratios / rankings
        for tag, ratio, metric in (('r100', 1.0, 'sort'), ('r50d', 0.5, 'degree'), ('r50s', 0.5, 'sort'), ('r30s', 0.3, 'sort')):
            torch.manual_seed(43)
            ig = build('IGCN', ds, dropout=0.3, feature_ratio=ratio, ranking_metric=metric)
            k = 'igcn_%s_' % tag
            g[k + 'adj_indices'], g[k + 'adj_values'] = _coo(ig.norm_adj)
            g[k + 'feat_indices'], g[k + 'feat_values_a0'] = _coo(ig.feat_mat)          # after update_feat_mat at alpha = 1
            g[k + 'feat_shape'] = np.array(ig.feat_mat.shape)
            g[k + 'row_sum'] = ig.row_sum.numpy().copy()
            g[k + 'user_map_k'], g[k + 'user_map_v'] = _map_arrays(ig.user_map)
            g[k + 'item_map_k'], g[k + 'item_map_v'] = _map_arrays(ig.item_map)
            g[k + 'emb_shape'] = np.array(ig.embedding.weight.shape)
            g[k + 'w'] = ig.w.detach().numpy().copy()
            g[k + 'state_keys'] = np.array(list(ig.state_dict().keys()))
            # dropout_sp_mat (model.py:263-275): identity in eval mode; in train mode keep iff floor(1 - p + U) == 1
            ig.eval()
            g[k + 'dropout_eval_is_identity'] = np.bool_(ref_model.NGCF.dropout_sp_mat(ig, ig.feat_mat) is ig.feat_mat)
            ig.train()
            torch.manual_seed(51)
            dropped = ref_model.NGCF.dropout_sp_mat(ig, ig.feat_mat)
            torch.manual_seed(51)
            g[k + 'dropout_rand'] = torch.rand(ig.feat_mat._nnz()).numpy().copy()      # the draw of model.py:267, same seed
            g[k + 'dropout_p'] = np.float64(ig.dropout)
            g[k + 'dropout_indices'], g[k + 'dropout_values'] = _coo(dropped)
            # bpr_forward / predict with get_rep replaced (model.py:293-299 via :448-449; :118-123 via :451-452)
            ig.get_rep = lambda: rep
            out = ig.bpr_forward(users, pos, neg)
            for t_tag, t in zip(('u', 'p', 'n', 'l2'), out):
                g[k + 'bpr_' + t_tag] = t.detach().numpy().copy()
            g[k + 'predict'] = ig.predict(pred_users).detach().numpy().copy()
            # anneal (model.py:374-381)
            for _ in range(3):
                ig.feat_mat_anneal()
            g[k + 'alpha_a3'] = np.float64(ig.alpha)
            idx3, g[k + 'feat_values_a3'] = _coo(ig.feat_mat)
            assert np.array_equal(idx3, g[k + 'feat_indices'])
            # save / load (model.py:454-466): keys of the checkpoint, and the state a fresh model has after load()
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                path = os.path.join(tmp, 'igcn.pth')
                ig.save(path)
                params = torch.load(path, map_location='cpu', weights_only=False)
                g[k + 'ckpt_keys'] = np.array(list(params.keys()))
                g[k + 'ckpt_state_keys'] = np.array(list(params['sate_dict'].keys()))
                torch.manual_seed(44)
                fresh = build('IGCN', ds, dropout=0.3, feature_ratio=ratio, ranking_metric='degree' if metric == 'sort' else 'sort')
                fresh.load(path)
            g[k + 'loaded_alpha'] = np.float64(fresh.alpha)
            g[k + 'loaded_feat_indices'], g[k + 'loaded_feat_values'] = _coo(fresh.feat_mat)
            g[k + 'loaded_row_sum'] = fresh.row_sum.numpy().copy()
            g[k + 'loaded_emb_equal'] = np.bool_(torch.equal(fresh.embedding.weight, ig.embedding.weight))

        # ---- the live-model update of the inductive scripts (run/dropui/igcn_dropui.py:26-32, run/dropit/igcn_dropit.py:
        #      26-35): a model built on the reduced split gets the full graph, generate_feat(is_updating=True)
        for split in ('dropui', 'dropit', 'dropui_half'):
            small = dataset_at(os.path.join(OUT, toy + '_' + split))
            for tag, ratio in (('r100', 1.0), ('r50', 0.5)):
                torch.manual_seed(45)
                ig = build('IGCN', small, dropout=0.3, feature_ratio=ratio, ranking_metric='sort')
                for _ in range(2):
                    ig.feat_mat_anneal()
                k = 'upd_%s_%s_' % (split, tag)
                g[k + 'small_n'] = np.array([small.n_users, small.n_items])
                g[k + 'small_feat_indices'], g[k + 'small_feat_values'] = _coo(ig.feat_mat)
                g[k + 'user_map_k'], g[k + 'user_map_v'] = _map_arrays(ig.user_map)
                g[k + 'item_map_k'], g[k + 'item_map_v'] = _map_arrays(ig.item_map)
                ig.config['dataset'] = ds
                ig.n_users, ig.n_items = ds.n_users, ds.n_items
                ig.norm_adj = ig.generate_graph(ds)
                ig.feat_mat, _, _, ig.row_sum = ig.generate_feat(ds, is_updating=True)
                ig.update_feat_mat()
                g[k + 'alpha'] = np.float64(ig.alpha)
                g[k + 'adj_indices'], g[k + 'adj_values'] = _coo(ig.norm_adj)
                g[k + 'feat_indices'], g[k + 'feat_values'] = _coo(ig.feat_mat)
                g[k + 'feat_shape'] = np.array(ig.feat_mat.shape)
                g[k + 'row_sum'] = ig.row_sum.numpy().copy()

        # ---- BASELINE config 1 END TO END with the reference's own classes (MF needs no DGL at all): get_model ->
        #      get_trainer -> BPRTrainer.train_one_epoch x 3 (DataLoader over BasicDataset.__getitem__, Adam) ->
        #      BasicTrainer.eval('test').  Recorded: initial tables, every batch the DataLoader produced, the epoch
        #      losses, the tables after training, the recommended ids and the metrics.
        import random as _random
        with quiet:
            ref_trainer = __import__('trainer')
        torch.manual_seed(61); _random.seed(61); np.random.seed(61)
        with quiet:
            mf2 = ref_model.get_model({'name': 'MF', 'embedding_size': 16, 'device': 'cpu'}, ds)
        tcfg = {'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': 1.e-2, 'l2_reg': 1.e-3, 'device': 'cpu', 'n_epochs': 3,
                'batch_size': 64, 'dataloader_num_workers': 0, 'test_batch_size': 7, 'topks': [5, 20] if ds.n_items > 20 else [5, 10]}
        with quiet:
            tr = ref_trainer.get_trainer(tcfg, ds, mf2)
        g['e2e_mf_user_emb0'] = mf2.user_embedding.weight.detach().numpy().copy()
        g['e2e_mf_item_emb0'] = mf2.item_embedding.weight.detach().numpy().copy()
        seen, loader = [], tr.dataloader

        class Tap:
            def __iter__(self_inner):
                for bd in loader:
                    seen.append(bd[:, 0, :].numpy().copy())
                    yield bd
        tr.dataloader = Tap()
        epoch_losses, per_epoch = [], []
        for _ in range(3):
            n0 = len(seen)
            epoch_losses.append(tr.train_one_epoch())
            per_epoch.append(len(seen) - n0)
        g['e2e_mf_batches'] = np.concatenate(seen, axis=0)
        g['e2e_mf_batch_sizes'] = np.array([b.shape[0] for b in seen], dtype=np.int64)
        g['e2e_mf_batches_per_epoch'] = np.array(per_epoch, dtype=np.int64)
        g['e2e_mf_epoch_losses'] = np.array(epoch_losses, dtype=np.float64)
        g['e2e_mf_lr'], g['e2e_mf_l2_reg'], g['e2e_mf_topks'] = tcfg['lr'], tcfg['l2_reg'], np.array(tcfg['topks'])
        g['e2e_mf_user_emb1'] = mf2.user_embedding.weight.detach().numpy().copy()
        g['e2e_mf_item_emb1'] = mf2.item_embedding.weight.detach().numpy().copy()
        got = {}
        orig_cm = tr.calculate_metrics

        def spy_cm(eval_data, rec_items, _got=got):
            _got['rec'] = rec_items.copy()
            return orig_cm(eval_data, rec_items)
        tr.calculate_metrics = spy_cm
        for stage in ('val', 'test'):
            _, metrics = tr.eval(stage)
            g['e2e_mf_%s_rec' % stage] = got['rec']
            for m in metrics:
                for kk in metrics[m]:
                    g['e2e_mf_%s_%s_%d' % (stage, m, kk)] = np.float64(metrics[m][kk])

        # ---- IMF (model.py:536-543) shares every DGL-free function with IGCN; record that it builds the same state
        torch.manual_seed(43)
        imf = build('IMF', ds, dropout=0.3, feature_ratio=0.5, ranking_metric='sort')
        fi, fv = _coo(imf.feat_mat)
        assert np.array_equal(fi, g['igcn_r50s_feat_indices']) and np.array_equal(fv, g['igcn_r50s_feat_values_a0'])
        g['imf_state_keys'] = np.array(list(imf.state_dict().keys()))

        np.savez_compressed(os.path.join(OUT, toy + '_model.npz'), **g)
        print('wrote', toy + '_model.npz', len(g), 'arrays')

    # the two gspmm callers really are out of reach, and nothing above went through the placeholder
    assert _InertPlaceholder.touched == [], _InertPlaceholder.touched
    try:
        lg2 = build('LightGCN', ds)
        lg2.get_rep()
        raise SystemExit('get_rep ran without DGL?')
    except AttributeError:
        assert _InertPlaceholder.touched == ['graph']
    print('dgl placeholder untouched by every recorded function; LightGCN.get_rep -> AttributeError(dgl.graph) as expected')


def main():
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import utils as ref_utils          # noqa: E402  (reference, unmodified)
        import dataset as ref_dataset      # noqa: E402
        import trainer as ref_trainer      # noqa: E402
    os.makedirs(OUT, exist_ok=True)
    quiet = contextlib.redirect_stdout(io.StringIO())
    import importlib.util

    def load_script(rel):                  # the dataset scripts live in directories without __init__.py
        spec = importlib.util.spec_from_file_location(rel.replace('/', '_')[:-3], os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    ref_dropui = load_script('run/dropui/dataset_dropui.py')
    ref_dropit = load_script('run/dropit/dataset_dropit.py')

    toys = {'toy_a': make_toy('toy_a', 30, 20, 11, quirks=False),
            'toy_b': make_toy('toy_b', 300, 200, 12, quirks=True)}

    for name, path in toys.items():
        g = {}
        with quiet:
            ds = ref_dataset.get_dataset({'name': 'ProcessedDataset', 'path': path, 'device': 'cpu'})
        g['n_users'], g['n_items'] = ds.n_users, ds.n_items
        g['train_array'] = np.array(ds.train_array, dtype=np.int64).reshape(-1, 2)
        g['len'] = len(ds)

        # utils.generate_daj_mat / get_sparse_tensor  (utils.py:32-49)
        adj = ref_utils.generate_daj_mat(ds)
        g['adj_indptr'], g['adj_indices'], g['adj_data'] = adj.indptr, adj.indices, adj.data
        g['adj_has_sorted_indices'] = adj.has_sorted_indices
        sp = ref_utils.get_sparse_tensor(adj, 'cpu')
        g['adj_coo_indices'] = sp.indices().numpy()
        g['adj_coo_values'] = sp.values().numpy()

        # utils.graph_rank_nodes  (utils.py:94-123)
        for metric in ('degree', 'sort'):
            ru, ri = ref_utils.graph_rank_nodes(ds, metric)
            g['rank_%s_users' % metric], g['rank_%s_items' % metric] = ru, ri

        # AuxiliaryDataset re-indexing with partial maps  (dataset.py:258-273)
        ru, ri = ref_utils.graph_rank_nodes(ds, 'degree')
        user_map = {int(u): j for j, u in enumerate(ru[:ds.n_users // 2])}
        item_map = {int(i): j for j, i in enumerate(ri[:ds.n_items // 2])}
        aux = ref_dataset.AuxiliaryDataset(ds, user_map, item_map)
        g['aux_user_keys'] = np.array(list(user_map.keys()), dtype=np.int64)
        g['aux_item_keys'] = np.array(list(item_map.keys()), dtype=np.int64)
        g['aux_len'] = len(aux)
        g['aux_rowlen'] = np.array([len(x) for x in aux.train_data], dtype=np.int64)
        g['aux_flat'] = np.array([i for x in aux.train_data for i in x], dtype=np.int64)

        # sampler semantics (dataset.py:119-131): shape/dtype and membership facts
        random.seed(5); np.random.seed(5)
        samples = np.stack([ds[0] for _ in range(400)], axis=0)
        g['sample_shape'] = np.array(samples.shape)
        g['samples'] = samples

        # BasicTrainer.eval + calculate_metrics  (trainer.py:109-177)
        model = RecordingModel(ds.n_users, ds.n_items, 8, seed=21)
        cfg = {'name': 'BasicTrainer', 'dataset': ds, 'model': model, 'topks': [5, 20] if ds.n_items > 20 else [5, 10],
               'device': 'cpu', 'n_epochs': 0, 'test_batch_size': 7}
        with quiet:
            tr = ref_trainer.BasicTrainer(cfg)
        g['eval_scores'] = model.scores.numpy()
        g['eval_score_u'], g['eval_score_i'] = model.score_u.numpy(), model.score_i.numpy()
        g['eval_topks'] = np.array(cfg['topks'])
        rec = {}
        orig = tr.calculate_metrics

        def spy(eval_data, rec_items, _rec=rec):
            _rec['items'] = rec_items.copy()
            return orig(eval_data, rec_items)
        tr.calculate_metrics = spy
        banned = np.arange(ds.n_items // 2, ds.n_items)
        for tag, stage, ban in (('train', 'train', None), ('val', 'val', None), ('test', 'test', None),
                                ('testban', 'test', banned)):
            _, metrics = tr.eval(stage, banned_items=ban)
            g['eval_%s_rec' % tag] = rec['items']
            for m in metrics:
                for k in metrics[m]:
                    g['eval_%s_%s_%d' % (tag, m, k)] = np.float64(metrics[m][k])
        g['eval_banned'] = banned

        # BasicTrainer.inductive_eval: the six masked evaluations (trainer.py:179-219)
        seq = []

        def spy2(eval_data, rec_items, _seq=seq):
            m = orig(eval_data, rec_items)
            _seq.append((rec_items.copy(), m))
            return m
        tr.calculate_metrics = spy2
        n_old_users, n_old_items = (2 * ds.n_users) // 3, (2 * ds.n_items) // 3
        with quiet:
            tr.inductive_eval(n_old_users, n_old_items)
        assert len(seq) == 6
        g['ind_n_old'] = np.array([n_old_users, n_old_items])
        for j, (rec_j, m_j) in enumerate(seq):
            g['ind_%d_rec' % j] = rec_j
            for m in m_j:
                for k in m_j[m]:
                    g['ind_%d_%s_%d' % (j, m, k)] = np.float64(m_j[m][k])
        tr.calculate_metrics = orig

        # calculate_metrics on hand-made inputs (k > |eval|, empty eval lists)
        rng = np.random.RandomState(3)
        hm_rec = rng.randint(0, ds.n_items, size=(ds.n_users, max(cfg['topks'])))
        metrics = orig(ds.test_data, hm_rec)
        g['hm_rec'] = hm_rec
        for m in metrics:
            for k in metrics[m]:
                g['hm_%s_%d' % (m, k)] = np.float64(metrics[m][k])

        # BPRTrainer.train_one_epoch: loss arithmetic + one Adam step (trainer.py:222-248)
        torch.manual_seed(7); random.seed(7); np.random.seed(7)
        model = RecordingModel(ds.n_users, ds.n_items, 8, seed=22)
        rep0 = model.rep.detach().numpy().copy()
        cfg = {'name': 'BPRTrainer', 'dataset': ds, 'model': model, 'topks': [5], 'device': 'cpu', 'n_epochs': 1,
               'test_batch_size': 7, 'optimizer': 'Adam', 'lr': 1e-2, 'l2_reg': 1e-2,
               'batch_size': len(ds) + 5, 'dataloader_num_workers': 0}
        with quiet:
            tr = ref_trainer.BPRTrainer(cfg)
        loss = tr.train_one_epoch()
        assert len(model.calls) == 1
        g['bpr_rep0'] = rep0
        g['bpr_users'], g['bpr_pos'], g['bpr_neg'] = (model.calls[0][k] for k in ('users', 'pos', 'neg'))
        g['bpr_loss'] = np.float64(loss)
        g['bpr_l2_reg'], g['bpr_lr'] = cfg['l2_reg'], cfg['lr']
        g['bpr_rep1'] = model.rep.detach().numpy().copy()

        # IGCNTrainer.train_one_epoch: main + auxiliary loss (trainer.py:281-320)
        torch.manual_seed(8); random.seed(8); np.random.seed(8)
        model = RecordingModel(ds.n_users, ds.n_items, 8, seed=23, template=True)
        rep0 = model.rep.detach().numpy().copy()
        emb0 = model.embedding.weight.detach().numpy().copy()
        w0 = model.w.detach().numpy().copy()
        cfg = {'name': 'IGCNTrainer', 'dataset': ds, 'model': model, 'topks': [5], 'device': 'cpu', 'n_epochs': 1,
               'test_batch_size': 7, 'optimizer': 'Adam', 'lr': 1e-2, 'l2_reg': 1e-3, 'aux_reg': 0.1,
               'batch_size': len(ds) + 5, 'dataloader_num_workers': 0}
        with quiet:
            tr = ref_trainer.IGCNTrainer(cfg)
        # record the auxiliary batch the reference draws (second DataLoader)
        aux_batches = []
        orig_aux = tr.aux_dataloader

        class Tap:
            def __iter__(self_inner):
                for b in orig_aux:
                    aux_batches.append(b.numpy().copy())
                    yield b
        tr.aux_dataloader = Tap()
        loss = tr.train_one_epoch()
        assert len(model.calls) == 1 and len(aux_batches) == 1 and model.anneal_calls == 1
        g['igcn_rep0'], g['igcn_emb0'], g['igcn_w0'] = rep0, emb0, w0
        g['igcn_users'], g['igcn_pos'], g['igcn_neg'] = (model.calls[0][k] for k in ('users', 'pos', 'neg'))
        g['igcn_aux'] = aux_batches[0][:, 0, :]
        g['igcn_loss'] = np.float64(loss)
        g['igcn_l2_reg'], g['igcn_aux_reg'], g['igcn_lr'] = cfg['l2_reg'], cfg['aux_reg'], cfg['lr']
        g['igcn_rep1'] = model.rep.detach().numpy().copy()
        g['igcn_emb1'] = model.embedding.weight.detach().numpy().copy()
        g['igcn_w1'] = model.w.detach().numpy().copy()

        # run/dropui/dataset_dropui.py:7-29 resize_dataset and run/dropit/dataset_dropit.py:6-9 dropit_dataset, each
        # followed by BasicDataset.output_dataset (dataset.py:40-44, :133-137): the three text files ARE the fixture
        for tag, fn, ratio in (('dropui', ref_dropui.resize_dataset, 0.8), ('dropit', ref_dropit.dropit_dataset, 0.8),
                               ('dropui_half', ref_dropui.resize_dataset, 0.5)):
            with quiet:
                fresh = ref_dataset.get_dataset({'name': 'ProcessedDataset', 'path': path, 'device': 'cpu'})
            fn(fresh, ratio)
            out_dir = os.path.join(OUT, name + '_' + tag)
            if os.path.isdir(out_dir):
                for f in os.listdir(out_dir):
                    os.remove(os.path.join(out_dir, f))
            else:
                os.makedirs(out_dir)
            fresh.output_dataset(out_dir)
            g['%s_n_users' % tag], g['%s_n_items' % tag] = fresh.n_users, fresh.n_items

        # BasicTrainer.train protocol (trainer.py:57-107) with scripted epochs: what is evaluated when, which
        # checkpoints are written / removed / re-loaded, when training stops, what is returned
        g.update(train_protocol_fixture(ref_trainer, ds))

        np.savez_compressed(os.path.join(OUT, name + '.npz'), **g)
        print('wrote', name, 'n_users', ds.n_users, 'n_items', ds.n_items, 'train pairs', len(ds))


if __name__ == '__main__':
    if '--model' in sys.argv[1:]:          # only the model.py fixtures (the others are left byte-identical)
        model_fixtures()
    else:
        main()
        model_fixtures()

"""The reference's OWN model.py against the oracle, the host builders and the HIP-side modules.

Fixtures: tests/golden/<toy>_model.npz, written by oracle/gen_golden.py:model_fixtures() from the
reference's model.py imported unmodified (its ``import dgl`` resolves to an inert object that has no
attributes and was asserted untouched): generate_graph (model.py:85-94), generate_feat incl.
is_updating (:386-421), update_feat_mat / feat_mat_anneal (:374-381), dropout_sp_mat (:263-275),
bpr_forward of MF / LightGCN / IGCN (:62-67, :108-116, :293-299) and predict (:69-72, :118-123) with
get_rep replaced by a recorded tensor, save / load (:454-466).  Only get_rep / inductive_rep_layer —
the gspmm callers — are not covered here (definitional, see oracle/oracle.py).

Tolerances: indices, maps, row sums, A_hat values, gathered rows: bit-exact.  pow() values: 1 ulp-ish
(rtol 2e-6: numpy / torch-CPU / device powf are different libm implementations).  Squared norms and
score blocks: rtol 2e-6 (summation order of 8..64 fp32 terms)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
F32 = np.float32
IGCN_TAGS = (('r100', 1.0, 'sort'), ('r50d', 0.5, 'degree'), ('r50s', 0.5, 'sort'), ('r30s', 0.3, 'sort'))
UPD_TAGS = [(s, t, r) for s in ('dropui', 'dropit', 'dropui_half') for t, r in (('r100', 1.0), ('r50', 0.5))]


@pytest.fixture(scope='module', params=['toy_a', 'toy_b'])
def mg(request):
    g = dict(np.load(os.path.join(GOLDEN, request.param + '_model.npz')))
    base = dict(np.load(os.path.join(GOLDEN, request.param + '.npz')))
    g.update(name=request.param, path=os.path.join(GOLDEN, request.param), train_array=base['train_array'],
             n_users=int(base['n_users']), n_items=int(base['n_items']))
    return g


def _maps(g, k):
    return ({int(a): int(b) for a, b in zip(g[k + 'user_map_k'], g[k + 'user_map_v'])},
            {int(a): int(b) for a, b in zip(g[k + 'item_map_k'], g[k + 'item_map_v'])})


def _coo_from_csr(rowptr, col):
    rowptr = np.asarray(rowptr)
    return np.stack([np.repeat(np.arange(rowptr.shape[0] - 1, dtype=np.int64), np.diff(rowptr)), np.asarray(col, dtype=np.int64)])


# ------------------------------------------------------------------------------------------------
# CPU: the oracle restatement and the host builders against the reference's outputs
# ------------------------------------------------------------------------------------------------
def test_fixture_was_made_without_touching_dgl():
    gen = os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), 'oracle', 'gen_golden.py')
    if not os.path.exists(gen):
        pytest.skip('the generator is not shipped to the GPU box (.gpurunignore)')
    src = open(gen).read()
    assert 'assert _InertPlaceholder.touched == []' in src
    assert 'def gspmm' not in src and 'def graph(' not in src             # nothing of DGL is implemented


def test_oracle_norm_adj_is_generate_graph(mg):
    """oracle.lightgcn_norm_adj == LightGCN.generate_graph (model.py:85-94), indices and float32 values bit for bit;
    IGCN reuses it (model.py:383-384)."""
    row, col, val = O.lightgcn_norm_adj(mg['train_array'], mg['n_users'], mg['n_items'])
    for k in ('lgcn_', 'igcn_r100_', 'igcn_r50d_'):
        np.testing.assert_array_equal(np.stack([row, col]), mg[k + 'adj_indices'])
        np.testing.assert_array_equal(val, mg[k + 'adj_values'])
    assert mg['lgcn_adj_values'].dtype == F32 and tuple(mg['lgcn_adj_shape']) == (mg['n_users'] + mg['n_items'],) * 2


def test_host_norm_adj_is_generate_graph(mg):
    from igcn_cf_amd.graph import normalized_adjacency_host
    rowptr, col, val = normalized_adjacency_host(mg['train_array'], mg['n_users'], mg['n_items'])
    np.testing.assert_array_equal(_coo_from_csr(rowptr, col), mg['lgcn_adj_indices'])
    np.testing.assert_array_equal(val, mg['lgcn_adj_values'])


@pytest.mark.parametrize('tag,ratio,metric', IGCN_TAGS)
def test_oracle_generate_feat_and_anneal(mg, tag, ratio, metric):
    """generate_feat (model.py:386-421): template maps from graph_rank_nodes, structure with duplicate pairs summed,
    row_sum; update_feat_mat (:374-377) REPLACES the summed value by row_sum^e; three anneals (:379-381)."""
    k = 'igcn_%s_' % tag
    nu, ni = mg['n_users'], mg['n_items']
    um, im = _maps(mg, k)
    if ratio < 1.:
        ru, ri = O.graph_rank_nodes(mg['train_array'], nu, ni, metric)
        assert um == {int(u): j for j, u in enumerate(ru[:int(nu * ratio)])}
        assert im == {int(i): j for j, i in enumerate(ri[:int(ni * ratio)])}
    else:
        assert um == {u: u for u in range(nu)} and im == {i: i for i in range(ni)}
    r, c, v, row_sum, _, _, shape = O.igcn_generate_feat(mg['train_array'], nu, ni, um, im)
    assert shape == tuple(mg[k + 'feat_shape']) and tuple(mg[k + 'emb_shape']) == (shape[1], 8)
    np.testing.assert_array_equal(np.stack([r, c]), mg[k + 'feat_indices'])
    np.testing.assert_array_equal(row_sum, mg[k + 'row_sum'])
    np.testing.assert_allclose(O.igcn_feat_values(r, row_sum, 1.), mg[k + 'feat_values_a0'], rtol=2e-6, atol=0)
    alpha = 1.
    for _ in range(3):
        alpha *= 0.99
    assert alpha == float(mg[k + 'alpha_a3'])
    np.testing.assert_allclose(O.igcn_feat_values(r, row_sum, alpha), mg[k + 'feat_values_a3'], rtol=2e-6, atol=0)
    assert np.array_equal(mg[k + 'w'], np.ones(8, dtype=F32))
    assert list(mg[k + 'state_keys']) == ['w', 'embedding.weight']
    # the host builder of the product
    from igcn_cf_amd.graph import feature_matrix_host
    rowptr, col, rs, shp = feature_matrix_host(mg['train_array'], nu, ni, None if ratio >= 1. else um, None if ratio >= 1. else im)
    np.testing.assert_array_equal(_coo_from_csr(rowptr, col), mg[k + 'feat_indices'])
    np.testing.assert_array_equal(rs, mg[k + 'row_sum'])
    assert shp == shape


@pytest.mark.parametrize('tag,ratio,metric', IGCN_TAGS)
def test_oracle_dropout_sp_mat(mg, tag, ratio, metric):
    """NGCF.dropout_sp_mat as IGCN uses it (model.py:263-275, :435): identity in eval mode; in train mode an edge is
    kept iff floor(1 - p + U) == 1 and a kept value is divided by (1 - p); structure = the kept edges in order."""
    k = 'igcn_%s_' % tag
    assert bool(mg[k + 'dropout_eval_is_identity'])
    p = float(mg[k + 'dropout_p'])
    keep = np.floor(F32(1. - p) + mg[k + 'dropout_rand']).astype(bool)
    np.testing.assert_array_equal(mg[k + 'feat_indices'][:, keep], mg[k + 'dropout_indices'])
    out = O.dropout_keep_scale(mg[k + 'feat_values_a0'], keep, p)
    np.testing.assert_array_equal(out[keep], mg[k + 'dropout_values'])
    assert 0.6 < keep.mean() < 0.8


def test_oracle_bpr_forward_and_predict(mg):
    """Which rows each model's L2 term reads: MF / LightGCN the RAW embedding rows (model.py:62-67, :108-116), IGCN
    the PROPAGATED rows (:293-299 via :448-449).  predict = rep[users] @ rep[n_users:].T (:118-123)."""
    nu = mg['n_users']
    users, pos, neg, rep = mg['users'], mg['pos'], mg['neg'], mg['rep']
    u, p, n, l2 = O.bpr_forward_mf(mg['mf_user_emb'], mg['mf_item_emb'], users, pos, neg)
    for a, tag in ((u, 'u'), (p, 'p'), (n, 'n')):
        np.testing.assert_array_equal(a, mg['mf_bpr_' + tag])
    np.testing.assert_allclose(l2, mg['mf_bpr_l2'], rtol=2e-6)
    np.testing.assert_allclose(mg['mf_user_emb'][mg['pred_users']] @ mg['mf_item_emb'].T, mg['mf_predict'], rtol=1e-5, atol=1e-7)

    u, p, n, l2 = O.bpr_forward_lightgcn(rep, mg['lgcn_emb'], nu, users, pos, neg)
    for a, tag in ((u, 'u'), (p, 'p'), (n, 'n')):
        np.testing.assert_array_equal(a, mg['lgcn_bpr_' + tag])
    np.testing.assert_allclose(l2, mg['lgcn_bpr_l2'], rtol=2e-6)
    # ... and NOT the propagated rows
    assert not np.allclose(O.bpr_forward_rep(rep, nu, users, pos, neg)[3], mg['lgcn_bpr_l2'], rtol=1e-3)
    np.testing.assert_allclose(O.predict(rep, nu, mg['pred_users']), mg['lgcn_predict'], rtol=1e-5, atol=1e-7)

    for tag, _, _ in IGCN_TAGS:
        k = 'igcn_%s_' % tag
        u, p, n, l2 = O.bpr_forward_rep(rep, nu, users, pos, neg)
        for a, t in ((u, 'u'), (p, 'p'), (n, 'n')):
            np.testing.assert_array_equal(a, mg[k + 'bpr_' + t])
        np.testing.assert_allclose(l2, mg[k + 'bpr_l2'], rtol=2e-6)
        np.testing.assert_allclose(O.predict(rep, nu, mg['pred_users']), mg[k + 'predict'], rtol=1e-5, atol=1e-7)
    assert list(mg['mf_state_keys']) == ['user_embedding.weight', 'item_embedding.weight']
    assert list(mg['lgcn_state_keys']) == ['embedding.weight'] and list(mg['imf_state_keys']) == ['w', 'embedding.weight']


@pytest.mark.parametrize('tag,ratio,metric', IGCN_TAGS)
def test_oracle_checkpoint_reload(mg, tag, ratio, metric):
    """IGCN.save / load (model.py:454-466): keys, and load() rebuilds the features from the LOADED maps
    (generate_feat(is_updating=True)) at the loaded alpha."""
    k = 'igcn_%s_' % tag
    assert list(mg[k + 'ckpt_keys']) == ['sate_dict', 'user_map', 'item_map', 'alpha']
    assert list(mg[k + 'ckpt_state_keys']) == ['w', 'embedding.weight']
    assert float(mg[k + 'loaded_alpha']) == float(mg[k + 'alpha_a3']) and bool(mg[k + 'loaded_emb_equal'])
    um, im = _maps(mg, k)
    r, c, _, row_sum, _, _, _ = O.igcn_generate_feat(mg['train_array'], mg['n_users'], mg['n_items'], um, im)
    np.testing.assert_array_equal(np.stack([r, c]), mg[k + 'loaded_feat_indices'])
    np.testing.assert_array_equal(row_sum, mg[k + 'loaded_row_sum'])
    np.testing.assert_array_equal(mg[k + 'loaded_feat_values'], mg[k + 'feat_values_a3'])


@pytest.mark.parametrize('split,tag,ratio', UPD_TAGS)
def test_oracle_live_update(mg, split, tag, ratio):
    """run/dropui/igcn_dropui.py:26-32, run/dropit/igcn_dropit.py:33-35: a model built on the reduced split gets the
    full graph; generate_feat(is_updating=True) keeps the OLD maps — new users / items have no template of their own
    and reach the old ones (and their global column) only."""
    k = 'upd_%s_%s_' % (split, tag)
    nu, ni = mg['n_users'], mg['n_items']
    um, im = _maps(mg, k)
    small_nu, small_ni = (int(x) for x in mg[k + 'small_n'])
    assert len(um) == int(small_nu * ratio) and len(im) == int(small_ni * ratio)
    r, c, _, row_sum, _, _, shape = O.igcn_generate_feat(mg['train_array'], nu, ni, um, im)
    assert shape == tuple(mg[k + 'feat_shape'])
    np.testing.assert_array_equal(np.stack([r, c]), mg[k + 'feat_indices'])
    np.testing.assert_array_equal(row_sum, mg[k + 'row_sum'])
    assert float(mg[k + 'alpha']) == 0.99 * 0.99
    np.testing.assert_allclose(O.igcn_feat_values(r, row_sum, float(mg[k + 'alpha'])), mg[k + 'feat_values'], rtol=2e-6, atol=0)
    row, col, val = O.lightgcn_norm_adj(mg['train_array'], nu, ni)
    np.testing.assert_array_equal(np.stack([row, col]), mg[k + 'adj_indices'])
    np.testing.assert_array_equal(val, mg[k + 'adj_values'])
    from igcn_cf_amd.graph import feature_matrix_host
    rowptr, colh, rs, shp = feature_matrix_host(mg['train_array'], nu, ni, um, im)
    np.testing.assert_array_equal(_coo_from_csr(rowptr, colh), mg[k + 'feat_indices'])
    np.testing.assert_array_equal(rs, mg[k + 'row_sum'])


# ------------------------------------------------------------------------------------------------
# GPU: the product's modules (device builders, HIP kernels) against the same reference outputs
# ------------------------------------------------------------------------------------------------
def _dataset(path):
    from igcn_cf_amd.dataset import get_dataset
    return get_dataset({'name': 'ProcessedDataset', 'path': path, 'device': 'cuda'})


def _csr_coo(csr):
    return _coo_from_csr(csr.rowptr.cpu().numpy(), csr.col.cpu().numpy())


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_hip_lightgcn_and_mf_against_reference_model(mg):
    from igcn_cf_amd.model import get_model
    ds = _dataset(mg['path'])
    nu = ds.n_users
    lg = get_model({'name': 'LightGCN', 'embedding_size': 8, 'n_layers': 2, 'device': 'cuda'}, ds)
    np.testing.assert_array_equal(_csr_coo(lg.norm_adj), mg['lgcn_adj_indices'])
    np.testing.assert_array_equal(lg.norm_adj.val.cpu().numpy(), mg['lgcn_adj_values'])
    assert list(lg.state_dict().keys()) == list(mg['lgcn_state_keys'])
    users, pos, neg = _t(mg['users']), _t(mg['pos']), _t(mg['neg'])
    rep = _t(mg['rep'])
    with torch.no_grad():
        lg.embedding.weight.copy_(_t(mg['lgcn_emb']))
    lg.get_rep = lambda needed_rows=None: rep                 # as in the generator: get_rep replaced by the recorded tensor
    out = lg.bpr_forward(users, pos, neg)
    for t, tag in zip(out[:3], ('u', 'p', 'n')):
        np.testing.assert_array_equal(t.detach().cpu().numpy(), mg['lgcn_bpr_' + tag])
    np.testing.assert_allclose(out[3].detach().cpu().numpy(), mg['lgcn_bpr_l2'], rtol=2e-6)
    np.testing.assert_allclose(lg.predict(_t(mg['pred_users'])).detach().cpu().numpy(), mg['lgcn_predict'], rtol=1e-5, atol=1e-7)
    # the fused loss terms the trainer uses read the same rows: mean l2 == mean of the reference's l2_norm_sq
    del lg.get_rep
    lg.train()
    terms = lg.bpr_loss_terms(users, pos, neg)
    assert abs(terms[1].item() - float(np.mean(mg['lgcn_bpr_l2'], dtype=np.float64))) < 2e-6 * max(1., abs(terms[1].item()))

    mf = get_model({'name': 'MF', 'embedding_size': 8, 'device': 'cuda'}, ds)
    assert list(mf.state_dict().keys()) == list(mg['mf_state_keys'])
    with torch.no_grad():
        mf.user_embedding.weight.copy_(_t(mg['mf_user_emb'])); mf.item_embedding.weight.copy_(_t(mg['mf_item_emb']))
    out = mf.bpr_forward(users, pos, neg)
    for t, tag in zip(out[:3], ('u', 'p', 'n')):
        np.testing.assert_array_equal(t.detach().cpu().numpy(), mg['mf_bpr_' + tag])
    np.testing.assert_allclose(out[3].detach().cpu().numpy(), mg['mf_bpr_l2'], rtol=2e-6)
    np.testing.assert_allclose(mf.predict(_t(mg['pred_users'])).detach().cpu().numpy(), mg['mf_predict'], rtol=1e-5, atol=1e-7)
    mf.train()
    terms = mf.bpr_loss_terms(users, pos, neg)
    assert abs(terms[1].item() - float(np.mean(mg['mf_bpr_l2'], dtype=np.float64))) < 2e-6 * max(1., abs(terms[1].item()))
    # fused scorer == reference predict -> topk on the recorded tables
    from igcn_cf_amd import ops
    k = 5
    idx, val = ops.score_topk(mf.user_embedding.weight.detach(), mf.item_embedding.weight.detach(), k, user_ids=_t(mg['pred_users']))
    ref_sorted = np.sort(mg['mf_predict'], axis=1)[:, ::-1][:, :k]
    np.testing.assert_allclose(val.cpu().numpy(), ref_sorted, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('tag,ratio,metric', IGCN_TAGS)
def test_hip_igcn_against_reference_model(mg, tag, ratio, metric, tmp_path):
    from igcn_cf_amd.model import get_model
    k = 'igcn_%s_' % tag
    ds = _dataset(mg['path'])
    cfg = {'name': 'IGCN', 'embedding_size': 8, 'n_layers': 2, 'device': 'cuda', 'dropout': 0.3, 'feature_ratio': ratio,
           'ranking_metric': metric}
    ig = get_model(cfg, ds)
    um, im = _maps(mg, k)
    assert ig.user_map == um and ig.item_map == im
    assert list(ig.user_map.keys()) == [int(x) for x in mg[k + 'user_map_k']]       # insertion order too (checkpoints)
    np.testing.assert_array_equal(_csr_coo(ig.norm_adj), mg[k + 'adj_indices'])
    np.testing.assert_array_equal(ig.norm_adj.val.cpu().numpy(), mg[k + 'adj_values'])
    assert ig.feat_mat.shape == tuple(mg[k + 'feat_shape']) and tuple(ig.embedding.weight.shape) == tuple(mg[k + 'emb_shape'])
    np.testing.assert_array_equal(_csr_coo(ig.feat_mat), mg[k + 'feat_indices'])
    np.testing.assert_array_equal(ig.row_sum.cpu().numpy(), mg[k + 'row_sum'])
    np.testing.assert_allclose(ig.feat_values().cpu().numpy(), mg[k + 'feat_values_a0'], rtol=2e-6, atol=0)
    assert list(ig.state_dict().keys()) == list(mg[k + 'state_keys'])
    assert torch.equal(ig.w.detach().cpu(), torch.ones(8))
    # eval-mode dropout is the identity: two eval get_rep calls agree and equal the dropout = 0 training pass
    ig.eval()
    with torch.no_grad():
        r_eval = ig.get_rep().clone()
    ig.train(); ig.dropout = 0.
    assert torch.equal(ig.get_rep().detach(), r_eval)
    ig.dropout = 0.3
    # bpr_forward / predict with get_rep replaced by the recorded tensor
    users, pos, neg, rep = _t(mg['users']), _t(mg['pos']), _t(mg['neg']), _t(mg['rep'])
    ig.get_rep = lambda needed_rows=None: rep
    out = ig.bpr_forward(users, pos, neg)
    for t, tg in zip(out[:3], ('u', 'p', 'n')):
        np.testing.assert_array_equal(t.detach().cpu().numpy(), mg[k + 'bpr_' + tg])
    np.testing.assert_allclose(out[3].detach().cpu().numpy(), mg[k + 'bpr_l2'], rtol=2e-6)
    np.testing.assert_allclose(ig.predict(_t(mg['pred_users'])).detach().cpu().numpy(), mg[k + 'predict'], rtol=1e-5, atol=1e-7)
    del ig.get_rep
    # anneal
    for _ in range(3):
        ig.feat_mat_anneal()
    assert ig.alpha == float(mg[k + 'alpha_a3'])
    np.testing.assert_allclose(ig.feat_values().cpu().numpy(), mg[k + 'feat_values_a3'], rtol=2e-6, atol=0)
    # save -> fresh model (other ranking) -> load: the reference's checkpoint layout and post-load state
    path = str(tmp_path / 'igcn.pth')
    ig.save(path)
    params = torch.load(path, map_location='cpu', weights_only=False)
    assert list(params.keys()) == list(mg[k + 'ckpt_keys']) and list(params['sate_dict'].keys()) == list(mg[k + 'ckpt_state_keys'])
    cfg2 = dict(cfg, ranking_metric='degree' if metric == 'sort' else 'sort')
    fresh = get_model(cfg2, ds)
    fresh.load(path)
    assert fresh.alpha == float(mg[k + 'loaded_alpha']) and torch.equal(fresh.embedding.weight, ig.embedding.weight)
    np.testing.assert_array_equal(_csr_coo(fresh.feat_mat), mg[k + 'loaded_feat_indices'])
    np.testing.assert_array_equal(fresh.row_sum.cpu().numpy(), mg[k + 'loaded_row_sum'])
    np.testing.assert_allclose(fresh.feat_values().cpu().numpy(), mg[k + 'loaded_feat_values'], rtol=2e-6, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize('split,tag,ratio', UPD_TAGS)
def test_hip_live_update_against_reference_model(mg, split, tag, ratio):
    """The mutation protocol of run/dropui/igcn_dropui.py:26-32 on the HIP-side IGCN, against what the reference's
    IGCN holds after the same assignments."""
    from igcn_cf_amd.model import get_model
    k = 'upd_%s_%s_' % (split, tag)
    small = _dataset(mg['path'] + '_' + split)
    full = _dataset(mg['path'])
    assert [small.n_users, small.n_items] == [int(x) for x in mg[k + 'small_n']]
    ig = get_model({'name': 'IGCN', 'embedding_size': 8, 'n_layers': 2, 'device': 'cuda', 'dropout': 0.3, 'feature_ratio': ratio,
                    'ranking_metric': 'sort'}, small)
    for _ in range(2):
        ig.feat_mat_anneal()
    np.testing.assert_array_equal(_csr_coo(ig.feat_mat), mg[k + 'small_feat_indices'])
    np.testing.assert_allclose(ig.feat_values().cpu().numpy(), mg[k + 'small_feat_values'], rtol=2e-6, atol=0)
    assert (ig.user_map, ig.item_map) == _maps(mg, k)
    ig.config['dataset'] = full
    ig.n_users, ig.n_items = full.n_users, full.n_items
    ig.norm_adj = ig.generate_graph(full)
    ig.feat_mat, _, _, ig.row_sum = ig.generate_feat(full, is_updating=True)
    ig.update_feat_mat()
    assert ig.alpha == float(mg[k + 'alpha'])
    np.testing.assert_array_equal(_csr_coo(ig.norm_adj), mg[k + 'adj_indices'])
    np.testing.assert_array_equal(ig.norm_adj.val.cpu().numpy(), mg[k + 'adj_values'])
    assert ig.feat_mat.shape == tuple(mg[k + 'feat_shape'])
    np.testing.assert_array_equal(_csr_coo(ig.feat_mat), mg[k + 'feat_indices'])
    np.testing.assert_array_equal(ig.row_sum.cpu().numpy(), mg[k + 'row_sum'])
    np.testing.assert_allclose(ig.feat_values().cpu().numpy(), mg[k + 'feat_values'], rtol=2e-6, atol=0)
    # and the updated model evaluates (new users / items included) without touching the template table's shape
    ig.eval()
    with torch.no_grad():
        rep = ig.get_rep()
    assert rep.shape == (full.n_users + full.n_items, 8) and torch.isfinite(rep).all()


# ------------------------------------------------------------------------------------------------
# BASELINE config 1 end to end: the reference's own MF + BPRTrainer + eval, run by the generator
# ------------------------------------------------------------------------------------------------
def _e2e_batches(mg):
    sizes = mg['e2e_mf_batch_sizes']
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    batches = [mg['e2e_mf_batches'][a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    per_epoch = mg['e2e_mf_batches_per_epoch']
    ecuts = np.concatenate([[0], np.cumsum(per_epoch)])
    return [batches[a:b] for a, b in zip(ecuts[:-1], ecuts[1:])]


def test_reference_mf_run_is_reproduced_by_the_restated_algorithm(mg):
    """The recorded run (model.py:52-72, trainer.py:222-248, :140-177 executed by the reference itself) against the
    oracle's loss arithmetic + torch Adam in float64 on the recorded batches: epoch losses (AverageMeter weighting,
    trainer.py:247), tables after three epochs, recommended ids and metrics."""
    u = torch.nn.Parameter(torch.from_numpy(mg['e2e_mf_user_emb0']).double())
    i = torch.nn.Parameter(torch.from_numpy(mg['e2e_mf_item_emb0']).double())
    opt = torch.optim.Adam([u, i], lr=float(mg['e2e_mf_lr']))
    l2_reg = float(mg['e2e_mf_l2_reg'])
    for ep, batches in enumerate(_e2e_batches(mg)):
        tot, cnt = 0., 0
        for b in batches:
            bt = torch.from_numpy(b)
            ue, pe, ne = u[bt[:, 0]], i[bt[:, 1]], i[bt[:, 2]]
            l2 = (ue ** 2).sum(1) + (pe ** 2).sum(1) + (ne ** 2).sum(1)
            loss = torch.nn.functional.softplus((ue * ne).sum(1) - (ue * pe).sum(1)).mean() + l2_reg * l2.mean()
            opt.zero_grad(); loss.backward(); opt.step()
            tot += float(loss.detach()) * len(b); cnt += len(b)
        assert abs(tot / cnt - float(mg['e2e_mf_epoch_losses'][ep])) < 2e-6
        for b in batches:                                          # sampler semantics of the recorded draws (dataset.py:119-131)
            assert b.shape[1] == 3 and (b[:, 2] >= 0).all()
    np.testing.assert_allclose(u.detach().numpy(), mg['e2e_mf_user_emb1'], atol=2e-5)
    np.testing.assert_allclose(i.detach().numpy(), mg['e2e_mf_item_emb1'], atol=2e-5)
    # evaluation of the REFERENCE's trained tables by the oracle: ids and metrics
    lists = {}
    for name in ('train', 'val', 'test'):
        lists[name], _ = O.read_data(os.path.join(mg['path'], name + '.txt'))
    scores = mg['e2e_mf_user_emb1'] @ mg['e2e_mf_item_emb1'].T
    kmax = int(mg['e2e_mf_topks'].max())
    for stage in ('val', 'test'):
        ex = [lists['train'][uu] + (lists['val'][uu] if stage == 'test' else []) for uu in range(mg['n_users'])]
        rec = O.eval_topk(scores, ex, None, k=kmax)
        ref = mg['e2e_mf_%s_rec' % stage]
        same = (rec == ref).all(axis=1)
        for uu in np.flatnonzero(~same):                           # only where scores tie to fp32 rounding
            np.testing.assert_allclose(np.sort(scores[uu, rec[uu]]), np.sort(scores[uu, ref[uu]]), rtol=1e-5)
        assert same.mean() > 0.98
        m = O.calculate_metrics(lists[stage], ref, [int(k) for k in mg['e2e_mf_topks']])
        for name in m:
            for k in m[name]:
                assert m[name][k] == mg['e2e_mf_%s_%s_%d' % (stage, name, k)]


@pytest.mark.gpu
def test_hip_mf_training_and_evaluation_reproduce_the_reference_run(mg):
    """BASELINE config 1 (MF + BPRTrainer, config.py:12-18) through the product path — fused loss node, captured steps for
    the full-size batches, fused Adam, fused score / mask / top-k — on the batches the reference's DataLoader produced:
    epoch losses <= 2e-6, tables after three epochs <= 2e-5, recommended ids and metrics equal to the reference's."""
    from igcn_cf_amd.model import get_model
    from igcn_cf_amd.trainer import get_trainer
    ds = _dataset(mg['path'])
    topks = [int(k) for k in mg['e2e_mf_topks']]
    model = get_model({'name': 'MF', 'embedding_size': 16, 'device': 'cuda'}, ds)
    with torch.no_grad():
        model.user_embedding.weight.copy_(_t(mg['e2e_mf_user_emb0'])); model.item_embedding.weight.copy_(_t(mg['e2e_mf_item_emb0']))
    trainer = get_trainer({'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': float(mg['e2e_mf_lr']), 'l2_reg': float(mg['e2e_mf_l2_reg']),
                           'device': 'cuda', 'n_epochs': 3, 'batch_size': 64, 'dataloader_num_workers': 0, 'test_batch_size': 7,
                           'topks': topks, 'host_metrics': True}, ds, model)
    model.train()
    for ep, batches in enumerate(_e2e_batches(mg)):
        tot, cnt = 0., 0
        for b in batches:
            loss = trainer.bpr_step(_t(b))
            tot += float(loss) * len(b); cnt += len(b)
        assert abs(tot / cnt - float(mg['e2e_mf_epoch_losses'][ep])) < 2e-6 * max(1., abs(tot / cnt))
    assert trainer._graph is not None                                  # the 64-triplet batches ran as captured graphs
    np.testing.assert_allclose(model.user_embedding.weight.detach().cpu().numpy(), mg['e2e_mf_user_emb1'], atol=2e-5)
    np.testing.assert_allclose(model.item_embedding.weight.detach().cpu().numpy(), mg['e2e_mf_item_emb1'], atol=2e-5)
    # evaluate the reference's trained tables (so that rounding of the training does not enter): ids and metrics
    with torch.no_grad():
        model.user_embedding.weight.copy_(_t(mg['e2e_mf_user_emb1'])); model.item_embedding.weight.copy_(_t(mg['e2e_mf_item_emb1']))
    scores = mg['e2e_mf_user_emb1'].astype(np.float64) @ mg['e2e_mf_item_emb1'].astype(np.float64).T
    for stage in ('val', 'test'):
        _, metrics = trainer.eval(stage)
        rec = trainer.last_rec_items.cpu().numpy()
        ref = mg['e2e_mf_%s_rec' % stage]
        same = (rec == ref).all(axis=1)
        for uu in np.flatnonzero(~same):
            np.testing.assert_allclose(np.sort(scores[uu, rec[uu]]), np.sort(scores[uu, ref[uu]]), rtol=1e-5)
        assert same.mean() > 0.98
        if same.all():
            for name in metrics:
                for k in metrics[name]:
                    assert metrics[name][k] == mg['e2e_mf_%s_%s_%d' % (stage, name, k)], (stage, name, k)
        else:
            for name in metrics:
                for k in metrics[name]:
                    assert abs(float(metrics[name][k]) - float(mg['e2e_mf_%s_%s_%d' % (stage, name, k)])) < 1e-3

"""The CPU oracle against the golden vectors the reference itself produced
(oracle/gen_golden.py: reference utils.py / dataset.py / trainer.py, unmodified)."""
import os

import numpy as np
import scipy.sparse as sp

from oracle import oracle as O


def _dataset(g):
    train, n1 = O.read_data(os.path.join(g['path'], 'train.txt'))
    val, n2 = O.read_data(os.path.join(g['path'], 'val.txt'))
    test, n3 = O.read_data(os.path.join(g['path'], 'test.txt'))
    return train, val, test, max(n1, n2, n3)


def test_reader_matches_reference(golden):
    train, val, test, n_items = _dataset(golden)
    assert len(train) == int(golden['n_users']) and n_items == int(golden['n_items'])
    ta = np.array([[u, i] for u in range(len(train)) for i in train[u]], dtype=np.int64).reshape(-1, 2)
    np.testing.assert_array_equal(ta, golden['train_array'])
    assert int(golden['len']) == len(ta)


def test_adjacency_bit_exact(golden):
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    row, col, val = O.generate_adj(golden['train_array'], nu, ni)
    np.testing.assert_array_equal(np.stack([row, col]), golden['adj_coo_indices'])
    np.testing.assert_array_equal(val, golden['adj_coo_values'])
    # CSR view of the same thing (utils.py:46-48)
    indptr = np.zeros(nu + ni + 1, dtype=np.int64)
    np.add.at(indptr, row + 1, 1)
    np.testing.assert_array_equal(np.cumsum(indptr), golden['adj_indptr'])
    assert val.max() >= (2. if golden['name'] == 'toy_b' else 1.)   # duplicates summed


def test_norm_adj_matches_scipy_formulation(golden):
    """model.py:85-94 restated; cross-checked against the same scipy expression
    evaluated on the reference's own generate_daj_mat output."""
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    adj = sp.csr_matrix((golden['adj_data'], golden['adj_indices'], golden['adj_indptr']), shape=(nu + ni, nu + ni))
    degree = np.maximum(1., np.array(np.sum(adj, axis=1)).squeeze())
    d_mat = sp.diags(np.power(degree, -0.5), format='csr', dtype=np.float32)
    ref = d_mat.dot(adj).dot(d_mat).tocoo()
    row, col, val = O.lightgcn_norm_adj(golden['train_array'], nu, ni)
    np.testing.assert_array_equal(row, ref.row)
    np.testing.assert_array_equal(col, ref.col)
    np.testing.assert_array_equal(val, ref.data.astype(np.float32))
    # symmetric
    m = sp.coo_matrix((val, (row, col)), shape=(nu + ni, nu + ni)).tocsr()
    assert abs(m - m.T).max() == 0.


def test_rank_nodes(golden):
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    for metric in ('degree', 'sort'):
        ru, ri = O.graph_rank_nodes(golden['train_array'], nu, ni, metric)
        np.testing.assert_array_equal(ru, golden['rank_%s_users' % metric])
        np.testing.assert_array_equal(ri, golden['rank_%s_items' % metric])


def test_auxiliary_reindex(golden):
    train, _, _, _ = _dataset(golden)
    user_map = {int(u): j for j, u in enumerate(golden['aux_user_keys'])}
    item_map = {int(i): j for j, i in enumerate(golden['aux_item_keys'])}
    out = O.auxiliary_train_data(train, user_map, item_map)
    np.testing.assert_array_equal(np.array([len(x) for x in out]), golden['aux_rowlen'])
    np.testing.assert_array_equal(np.array([i for x in out for i in x], dtype=np.int64), golden['aux_flat'])
    assert int(golden['aux_len']) == int(golden['len'])


def test_sampler_semantics(golden):
    """dataset.py:119-131: [neg_ratio, 3] int64, positive in train list, negative not."""
    train, _, _, n_items = _dataset(golden)
    s = golden['samples']
    assert s.shape[1:] == (1, 3) and s.dtype == np.int64
    for u, p, n in s[:, 0, :]:
        assert train[u] and p in train[u] and n not in train[u] and 0 <= n < n_items


def uniq_all(s, picks):
    return all(np.sum(s == v) == 1 for v in s[picks])


def _excl(train, val, stage):
    if stage == 'train':
        return None
    return [train[u] + (val[u] if stage == 'test' else []) for u in range(len(train))]


def test_eval_topk_and_metrics(golden):
    train, val, test, _ = _dataset(golden)
    topks = [int(k) for k in golden['eval_topks']]
    data = {'train': train, 'val': val, 'test': test}
    for tag, stage, ban in (('train', 'train', None), ('val', 'val', None), ('test', 'test', None),
                            ('testban', 'test', golden['eval_banned'])):
        rec = O.eval_topk(golden['eval_scores'], _excl(train, val, stage), ban, k=max(topks))
        ref = golden['eval_%s_rec' % tag]
        scores = np.array(golden['eval_scores'], copy=True)
        # identical ids wherever the reference's choice is not a tie / a masked (-inf) slot
        ex = _excl(train, val, stage)
        all_finite = True
        for u in range(rec.shape[0]):
            s = scores[u].copy()
            if ex is not None and len(ex[u]):
                s[np.asarray(ex[u])] = -np.inf
            if ban is not None:
                s[ban] = -np.inf
            np.testing.assert_array_equal(s[rec[u]], s[ref[u]])          # same score sequence
            finite = np.isfinite(s[ref[u]])
            all_finite &= bool(finite.all()) and bool(uniq_all(s, ref[u]))
            uniq = np.array([np.sum(s == v) == 1 for v in s[ref[u]]])
            np.testing.assert_array_equal(rec[u][finite & uniq], ref[u][finite & uniq])
        m = O.calculate_metrics(data[stage], ref, topks)
        for name in m:
            for k in m[name]:
                assert m[name][k] == golden['eval_%s_%s_%d' % (tag, name, k)]
        if not all_finite:      # ties / -inf fill-ins inside the top-k: torch.topk's pick among them is arbitrary
            continue
        m2 = O.calculate_metrics(data[stage], rec, topks)
        for name in m2:
            for k in m2[name]:
                assert abs(m2[name][k] - golden['eval_%s_%s_%d' % (tag, name, k)]) < 1e-3


def test_metrics_handmade(golden):
    _, _, test, _ = _dataset(golden)
    topks = [int(k) for k in golden['eval_topks']]
    m = O.calculate_metrics(test, golden['hm_rec'], topks)
    for name in m:
        for k in m[name]:
            assert m[name][k] == golden['hm_%s_%d' % (name, k)]


def test_bpr_loss_arithmetic(golden):
    nu = int(golden['n_users'])
    rep = golden['bpr_rep0']
    u, p, n, l2 = O.bpr_forward_rep(rep, nu, golden['bpr_users'], golden['bpr_pos'], golden['bpr_neg'])
    bpr, reg = O.bpr_loss(u, p, n, l2, float(golden['bpr_l2_reg']))
    assert abs(bpr + reg - float(golden['bpr_loss'])) < 2e-6
    # one Adam step from zero state moves each touched parameter by lr * sign(grad)
    gu, gp, gn = O.bpr_grads(u, p, n)
    grad = np.zeros(rep.shape, dtype=np.float64)
    np.add.at(grad, golden['bpr_users'], gu + 2 * float(golden['bpr_l2_reg']) / len(u) * u)
    np.add.at(grad, nu + golden['bpr_pos'], gp + 2 * float(golden['bpr_l2_reg']) / len(u) * p)
    np.add.at(grad, nu + golden['bpr_neg'], gn + 2 * float(golden['bpr_l2_reg']) / len(u) * n)
    step = golden['bpr_rep1'].astype(np.float64) - rep
    big = np.abs(grad) > 1e-6
    # torch Adam, step 1, zero state: p -= lr * g / (|g| + eps), eps = 1e-8
    np.testing.assert_allclose(step[big], -float(golden['bpr_lr']) * grad[big] / (np.abs(grad[big]) + 1e-8),
                               rtol=2e-3, atol=1e-6)
    assert np.all(step[grad == 0] == 0)


def test_igcn_aux_loss_arithmetic(golden):
    nu = int(golden['n_users'])
    rep, emb, w = golden['igcn_rep0'], golden['igcn_emb0'], golden['igcn_w0']
    u, p, n, l2 = O.bpr_forward_rep(rep, nu, golden['igcn_users'], golden['igcn_pos'], golden['igcn_neg'])
    bpr, reg = O.bpr_loss(u, p, n, l2, float(golden['igcn_l2_reg']))
    aux = golden['igcn_aux']
    n_tpl_users = nu                                       # len(model.user_map), trainer.py:307-308
    au, ap, an = emb[aux[:, 0]], emb[aux[:, 1] + n_tpl_users], emb[aux[:, 2] + n_tpl_users]
    aux_loss, _ = O.bpr_loss(au, ap, an, None, 0., w=w)
    total = bpr + reg + float(golden['igcn_aux_reg']) * aux_loss
    assert abs(total - float(golden['igcn_loss'])) < 2e-6


def test_spmm_restatement_matches_scipy_and_f64(golden):
    """The definitional SpMM (restated gspmm 'mul','sum'): Y = M @ X."""
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    row, col, val = O.lightgcn_norm_adj(golden['train_array'], nu, ni)
    rng = np.random.RandomState(0)
    x = (rng.randn(nu + ni, 16) * 0.1).astype(np.float32)
    y = O.spmm_coo(row, col, val, x)
    m = sp.coo_matrix((val, (row, col)), shape=(nu + ni, nu + ni)).tocsr()
    np.testing.assert_allclose(y, m @ x, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(y, O.spmm_coo_f64(row, col, val, x), rtol=1e-5, atol=1e-7)
    rep = O.lightgcn_get_rep((row, col, val), x, 3)
    m64 = m.astype(np.float64)
    x64 = x.astype(np.float64)
    ref = (x64 + m64 @ x64 + m64 @ (m64 @ x64) + m64 @ (m64 @ (m64 @ x64))) / 4
    np.testing.assert_allclose(rep, ref, rtol=1e-5, atol=1e-7)


def test_feat_matrix_structure(golden):
    """model.py:386-421 restated: F is N x (Tu+Ti+2), nnz = 2T' + N, row_sum = deg+1."""
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    r, c, v, row_sum, um, im, shape = O.igcn_generate_feat(golden['train_array'], nu, ni)
    assert shape == (nu + ni, nu + ni + 2)
    arow, acol, aval = O.generate_adj(golden['train_array'], nu, ni)
    deg = np.zeros(nu + ni, dtype=np.float32)
    np.add.at(deg, arow, aval)
    np.testing.assert_array_equal(row_sum, deg + 1)
    assert v.sum() == 2 * len(golden['train_array']) + nu + ni
    # alpha = 1 -> deg^-1/2 ; alpha -> 0 -> deg^-1
    np.testing.assert_allclose(O.igcn_feat_values(r, row_sum, 1.), row_sum[r] ** -0.5, rtol=1e-6)
    np.testing.assert_allclose(O.igcn_feat_values(r, row_sum, 0.), row_sum[r] ** -1.0, rtol=1e-6)

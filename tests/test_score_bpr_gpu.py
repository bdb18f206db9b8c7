"""Fused BPR scoring, fused score+mask+top-k, hit matrix and the device sampler
(all through the C ABI) against the CPU oracle and the golden fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _lists(g):
    out = {}
    for name in ('train', 'val', 'test'):
        out[name], _ = O.read_data(os.path.join(g['path'], name + '.txt'))
    return out


def _excl_csr(lists, stage):
    ex = [sorted(lists['train'][u] + (lists['val'][u] if stage == 'test' else [])) for u in range(len(lists['train']))]
    rowptr = np.zeros(len(ex) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in ex], out=rowptr[1:])
    col = np.array([i for x in ex for i in x], dtype=np.int32)
    return ex, rowptr, col


# ---------------------------------------------------------------- top-k --------------------------------
def _check_topk(scores, idx, val, ex_lists, banned, k):
    """idx/val from the kernel vs the oracle ordering on the dense masked scores."""
    s = np.array(scores, dtype=np.float32, copy=True)
    if ex_lists is not None:
        for u, items in enumerate(ex_lists):
            if len(items):
                s[u, np.asarray(items, dtype=np.int64)] = -np.inf
    if banned is not None:
        s[:, banned] = -np.inf
    ref = O.eval_topk(s, None, None, k=k)
    np.testing.assert_array_equal(idx, ref)
    np.testing.assert_array_equal(val, np.take_along_axis(s, ref, axis=1))


@pytest.mark.parametrize('d,n_users,n_items,k', [(64, 300, 1000, 20), (8, 70, 50, 20), (128, 257, 4500, 20),
                                                  (32, 33, 9000, 5), (64, 1000, 20000, 50), (16, 5, 64, 64),
                                                  (50, 100, 3000, 20), (6, 40, 200, 7),
                                                  (64, 130, 2000, 100), (32, 70, 1500, 200), (128, 40, 900, 64),
                                                  (64, 65, 700, 25), (16, 129, 333, 256), (64, 2500, 9000, 60),
                                                  (64, 200, 40, 20), (256, 70, 1500, 20), (192, 33, 800, 30),
                                                  # the two-stage path's upper end: k + 6 candidates up to k = 58, then 64 - k
                                                  (64, 400, 5000, 58), (64, 400, 5000, 59), (128, 130, 3000, 60), (64, 90, 4000, 61)])
def test_score_topk_matches_dense_oracle(d, n_users, n_items, k):
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(d + n_items)
    # small-integer embeddings: every dot product is exact in fp32 whatever the summation order,
    # so ids AND values must match the oracle exactly, ties included (lower id first)
    U = rng.integers(-3, 4, size=(n_users, d)).astype(np.float32)
    I = rng.integers(-3, 4, size=(n_items, d)).astype(np.float32)
    ex = [sorted(rng.choice(n_items, size=int(rng.integers(0, max(1, min(30, n_items - k + 1)))), replace=False).tolist())
          for _ in range(n_users)]
    ex[0] = []
    rowptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum([len(x) for x in ex], out=rowptr[1:])
    col = np.array([i for x in ex for i in x], dtype=np.int32)
    banned = np.sort(rng.choice(n_items, size=n_items // 7, replace=False))
    bmask = np.zeros(n_items, dtype=np.uint8); bmask[banned] = 1
    scores = U @ I.T
    users = np.arange(n_users, dtype=np.int64)
    idx, val = score_topk(_dev(U), _dev(I), k, user_ids=_dev(users), excl_rowptr=_dev(rowptr), excl_col=_dev(col),
                          banned=_dev(bmask))
    _check_topk(scores, idx.cpu().numpy(), val.cpu().numpy(), ex, banned, k)
    # no masks; a permuted user subset through user_ids
    sub = rng.permutation(n_users)[: max(1, n_users // 2)].astype(np.int64)
    idx, val = score_topk(_dev(U), _dev(I), k, user_ids=_dev(sub))
    _check_topk(scores[sub], idx.cpu().numpy(), val.cpu().numpy(), None, None, k)
    if d in (64, 128) and k <= 60:
        # the two-stage path (fp16 candidate sweep + exact fp32 re-scoring; d = 64 and 128): the same lists; integer scores tie in
        # droves, so many users here take its fall-back through the fp32 sweep
        idx, val = score_topk(_dev(U), _dev(I), k, user_ids=_dev(users), excl_rowptr=_dev(rowptr), excl_col=_dev(col),
                              banned=_dev(bmask), mode='fast')
        _check_topk(scores, idx.cpu().numpy(), val.cpu().numpy(), ex, banned, k)
        idx, val = score_topk(_dev(U), _dev(I), k, user_ids=_dev(sub), mode='fast')
        _check_topk(scores[sub], idx.cpu().numpy(), val.cpu().numpy(), None, None, k)


@pytest.mark.parametrize('mode', ['exact', 'fast', 'fast_64_user_groups', 'fast_no_sharing', 'fast_bf16', 'fast_f16x2', 'fast_d128', 'fast_d128_f16x2', 'fast_d128_f16x2_narrow'])
def test_score_topk_random_floats_match_sets(mode):
    """Gaussian fp32 embeddings: same top-k sets as the float64 ranking except where the
    k-th and (k+1)-th scores are within fp32 rounding of each other — for the fp32 sweep and for the two-stage
    path, whose lists must moreover be those of the fp32 sweep bit for bit."""
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(1)
    n_users, n_items, d, k = 512, 30000, (128 if 'd128' in mode else 64), 20
    U = (rng.standard_normal((n_users, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((n_items, d)) * 0.1).astype(np.float32)
    I[100] = I[7]; I[20000] = I[7]                       # identical item rows: exact ties, decided by the lower id
    U[5] = 0.0                                           # an all-zero user: every score ties at 0
    from igcn_cf_amd import _lib
    # candidate sweep: one fp16 plane each side (default) / two bf16 planes each side (d = 64) / one fp16 item plane and two
    # user planes — at d = 128 that one runs two user groups per wave at one wave per SIMD, or ('narrow') one group at two waves
    if mode == 'fast_bf16':
        _lib.set_tuning('topk_fast_mode', 1)
    if mode == 'fast_64_user_groups':                    # (a batch this small runs 32-user wave-groups by default)
        _lib.set_tuning('topk_fast_narrow', 0)
    if mode == 'fast_no_sharing':                        # (the pieces of this small batch's sweeps share thresholds by default)
        _lib.set_tuning('topk_fast_share', 0)
    if 'f16x2' in mode:
        _lib.set_tuning('topk_fast_mode', 2)
        _lib.set_tuning('topk_fast_wide', 0 if 'narrow' in mode else None)
    if mode.startswith('fast'):
        mode = 'fast'
    try:
        idx, val = score_topk(_dev(U), _dev(I), k, mode=mode)
        if mode == 'fast':
            idx_e, val_e = score_topk(_dev(U), _dev(I), k, mode='exact')
            assert torch.equal(idx, idx_e) and torch.equal(val, val_e)
            # users 1e-6 times smaller than the rest of the batch: their scaled elements are fp16 subnormals in the sweep
            U2 = U.copy(); U2[::7] *= np.float32(1e-6)
            a = score_topk(_dev(U2), _dev(I), k, mode='fast')
            b = score_topk(_dev(U2), _dev(I), k, mode='exact')
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            # badly scaled tables (fp16 has 5 exponent bits: the sweep rescales both tables by a power of two)
            for su, si in ((1e-6, 3e4), (2e5, 1e-7)):
                a = score_topk(_dev(U * np.float32(su)), _dev(I * np.float32(si)), k, mode='fast')
                b = score_topk(_dev(U * np.float32(su)), _dev(I * np.float32(si)), k, mode='exact')
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    finally:
        _lib.set_tuning('topk_fast_mode', None)
        _lib.set_tuning('topk_fast_wide', None)
        _lib.set_tuning('topk_fast_narrow', None)
        _lib.set_tuning('topk_fast_share', None)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    assert list(idx[5]) == list(range(k))
    s64 = U.astype(np.float64) @ I.astype(np.float64).T
    ref = np.argsort(-s64, axis=1, kind='stable')[:, :k + 1]
    np.testing.assert_allclose(val, np.take_along_axis(s64, idx, axis=1), rtol=1e-5, atol=1e-6)
    assert np.all(np.diff(val, axis=1) <= 0)
    bad = 0
    for u in range(n_users):
        if set(idx[u]) != set(ref[u, :k]):
            gap = s64[u, ref[u, k - 1]] - s64[u, ref[u, k]]
            assert gap < 1e-6, (u, gap)
            bad += 1
    assert bad <= 2


def _golden_masked_scores(golden, ex, ban):
    s = np.array(golden['eval_scores'], dtype=np.float32, copy=True)
    if ex is not None:
        for u, items in enumerate(ex):
            if len(items):
                s[u, np.asarray(items, dtype=np.int64)] = -np.inf
    if ban is not None:
        s[:, ban] = -np.inf
    return s


def _assert_same_ranking(idx, val, ref, s):
    """Kernel picks vs the reference's picks on masked scores `s`: identical ids wherever the
    reference's choice is not a tie / -inf fill-in / within fp32 rounding of its neighbours."""
    for u in range(ref.shape[0]):
        np.testing.assert_allclose(val[u], s[u][ref[u]], rtol=2e-6, atol=2e-6)
        fin = np.isfinite(s[u])
        for j in range(ref.shape[1]):
            v = s[u, ref[u, j]]
            if not np.isfinite(v):
                continue
            near = np.sum(np.abs(s[u][fin] - v) <= 2e-5 * max(1., abs(v)))
            if near == 1:
                assert idx[u, j] == ref[u, j], (u, j)


def test_eval_golden_topk(golden):
    """Reference BasicTrainer.eval outputs (trainer.py:140-164): the score matrix the reference
    ranked is a rank-16 product whose factors are in the fixture; the fused kernel gets the
    factors and must reproduce the recommended ids."""
    from igcn_cf_amd.ops import score_topk
    lists = _lists(golden)
    U, I = _dev(golden['eval_score_u']), _dev(golden['eval_score_i'])
    n_items = I.shape[0]
    k = int(max(golden['eval_topks']))
    for tag, stage, ban in (('train', 'train', None), ('val', 'val', None), ('test', 'test', None),
                            ('testban', 'test', golden['eval_banned'])):
        kw = {}
        ex = None
        if stage != 'train':
            ex, rowptr, col = _excl_csr(lists, stage)
            kw.update(excl_rowptr=_dev(rowptr), excl_col=_dev(col))
        if ban is not None:
            bm = np.zeros(n_items, dtype=np.uint8); bm[ban] = 1
            kw['banned'] = _dev(bm)
        idx, val = score_topk(U, I, k, **kw)
        _assert_same_ranking(idx.cpu().numpy(), val.cpu().numpy(), golden['eval_%s_rec' % tag],
                             _golden_masked_scores(golden, ex, ban))


class _FactorModel:
    """Scores = U . I^T from fixed factors, through the fused kernel (model.recommend contract)."""

    def __init__(self, U, I):
        self.U, self.I = U, I

    def eval(self):
        pass

    def recommend(self, users, k, excl_rowptr=None, excl_col=None, banned=None):
        from igcn_cf_amd.ops import score_topk
        return score_topk(self.U, self.I, k, user_ids=users, excl_rowptr=excl_rowptr, excl_col=excl_col, banned=banned)[0]


def test_inductive_eval_golden(golden, capsys):
    """The six masked evaluations of the reference's BasicTrainer.inductive_eval
    (trainer.py:179-219) on the same split and score factors: same metrics."""
    from igcn_cf_amd.dataset import get_dataset
    from igcn_cf_amd.trainer import BasicTrainer
    ds = get_dataset({'name': 'ProcessedDataset', 'path': golden['path'], 'device': 'cuda'})
    topks = [int(k) for k in golden['eval_topks']]
    model = _FactorModel(_dev(golden['eval_score_u']), _dev(golden['eval_score_i']))
    tr = BasicTrainer({'name': 'BasicTrainer', 'dataset': ds, 'model': model, 'topks': topks, 'device': 'cuda',
                       'n_epochs': 0, 'test_batch_size': 7})
    got = []
    orig = tr.eval

    def spy(stage, banned_items=None, _eval_lists=None):
        res = orig(stage, banned_items=banned_items, _eval_lists=_eval_lists)
        rp, cl = _eval_lists[0].cpu().numpy(), _eval_lists[1].cpu().numpy()      # the masked test lists of this variant
        got.append((tr.last_rec_items.cpu().numpy(), res[1], [cl[rp[u]:rp[u + 1]].tolist() for u in range(ds.n_users)]))
        return res
    tr.eval = spy
    test_before = [list(x) for x in ds.test_data]
    n_old_users, n_old_items = (int(v) for v in golden['ind_n_old'])
    tr.inductive_eval(n_old_users, n_old_items)
    assert len(got) == 6 and [list(x) for x in ds.test_data] == test_before      # the dataset's lists are never touched
    assert 'Old users and old items result.' in capsys.readouterr().out
    lists = _lists(golden)
    ex, _, _ = _excl_csr(lists, 'test')
    n_items = ds.n_items
    bans = [None, None, None, np.arange(n_old_items, n_items), np.arange(n_old_items), np.arange(n_old_items, n_items)]
    for j, (rec, metrics, eval_lists) in enumerate(got):
        ref = golden['ind_%d_rec' % j]
        s = _golden_masked_scores(golden, ex, bans[j])
        _assert_same_ranking(rec, np.take_along_axis(s, rec, axis=1), ref, s)
        # same masked test lists + the reference's own picks -> the reference's metrics, bit for bit
        # (ties inside a top-k are the only freedom between `rec` and `ref`)
        m = tr.calculate_metrics(eval_lists, ref)
        for name in m:
            for k in m[name]:
                refv = float(golden['ind_%d_%s_%d' % (j, name, k)])
                assert m[name][k] == refv or (np.isnan(refv) and np.isnan(m[name][k])), (j, name, k)
                if np.array_equal(rec, ref):                     # eval() reduces on the device: float32 rounding only
                    assert abs(metrics[name][k] - refv) < 1e-6 or np.isnan(refv)


def test_hit_matrix_and_metrics_golden(golden):
    """Reference calculate_metrics values (trainer.py:109-138) from the device hit matrix."""
    from igcn_cf_amd.trainer import BasicTrainer
    lists = _lists(golden)

    class DS:
        n_users, n_items = int(golden['n_users']), int(golden['n_items'])
    topks = [int(k) for k in golden['eval_topks']]
    tr = BasicTrainer({'name': 'BasicTrainer', 'dataset': DS(), 'model': None, 'topks': topks, 'device': 'cuda',
                       'n_epochs': 0, 'test_batch_size': 7})
    for tag, stage in (('train', 'train'), ('val', 'val'), ('test', 'test'), ('testban', 'test')):
        m = tr.calculate_metrics(lists[stage], golden['eval_%s_rec' % tag])
        for name in m:
            for k in m[name]:
                assert m[name][k] == golden['eval_%s_%s_%d' % (tag, name, k)], (tag, name, k)
    m = tr.calculate_metrics(lists['test'], golden['hm_rec'])
    for name in m:
        for k in m[name]:
            assert m[name][k] == golden['hm_%s_%d' % (name, k)]


# ---------------------------------------------------------------- BPR ----------------------------------
def test_bpr_golden_loss_and_adam_step(golden):
    """BPRTrainer.train_one_epoch of the reference (trainer.py:231-248) on its recorded batch:
    same loss, same parameters after one Adam step."""
    from igcn_cf_amd.ops import bpr_loss_terms
    nu = int(golden['n_users'])
    rep = torch.nn.Parameter(_dev(golden['bpr_rep0']))
    users, pos, neg = _dev(golden['bpr_users']), _dev(golden['bpr_pos']), _dev(golden['bpr_neg'])
    opt = torch.optim.Adam([rep], lr=float(golden['bpr_lr']))
    terms = bpr_loss_terms(rep, rep, rep, rep, None, users, pos, neg, nu, nu)
    loss = terms[0] + float(golden['bpr_l2_reg']) * terms[1]
    assert abs(loss.item() - float(golden['bpr_loss'])) < 2e-6
    opt.zero_grad(); loss.backward(); opt.step()
    # Adam's first step is lr * g / (|g| + eps): compare where the gradient is well above eps
    g = rep.grad.cpu().numpy()
    got, ref = rep.detach().cpu().numpy(), golden['bpr_rep1']
    big = np.abs(g) > 1e-5
    np.testing.assert_allclose(got[big], ref[big], rtol=0, atol=2e-6)
    assert np.abs(got - ref).max() < 2e-4            # tiny gradients: sign may flip under eps
    assert np.all(got[g == 0] == golden['bpr_rep0'][g == 0])


def test_igcn_golden_loss_and_adam_step(golden):
    """IGCNTrainer.train_one_epoch of the reference (trainer.py:294-320): main + auxiliary loss."""
    from igcn_cf_amd.ops import bpr_loss_terms
    nu = int(golden['n_users'])
    rep = torch.nn.Parameter(_dev(golden['igcn_rep0']))
    emb = torch.nn.Parameter(_dev(golden['igcn_emb0']))
    w = torch.nn.Parameter(_dev(golden['igcn_w0']))
    users, pos, neg = _dev(golden['igcn_users']), _dev(golden['igcn_pos']), _dev(golden['igcn_neg'])
    aux = golden['igcn_aux']
    au, ap, an = _dev(aux[:, 0]), _dev(aux[:, 1]), _dev(aux[:, 2])
    opt = torch.optim.Adam([rep, emb, w], lr=float(golden['igcn_lr']))
    terms = bpr_loss_terms(rep, rep, rep, rep, None, users, pos, neg, nu, nu)
    aux_loss = bpr_loss_terms(emb, emb, None, None, w, au, ap, an, nu, 0)[0]
    loss = terms[0] + float(golden['igcn_l2_reg']) * terms[1] + float(golden['igcn_aux_reg']) * aux_loss
    assert abs(loss.item() - float(golden['igcn_loss'])) < 2e-6
    opt.zero_grad(); loss.backward(); opt.step()
    for name, p in (('rep', rep), ('emb', emb), ('w', w)):
        g = p.grad.cpu().numpy()
        got, ref = p.detach().cpu().numpy(), golden['igcn_%s1' % name]
        big = np.abs(g) > 1e-5
        np.testing.assert_allclose(got[big], ref[big], rtol=0, atol=2e-6)
        assert np.abs(got - ref).max() < 2e-4


@pytest.mark.parametrize('d', [64, 128, 24])
def test_bpr_grads_against_oracle(d):
    """Separate user / item tables (MF layout, model.py:62-67), duplicates in the batch."""
    from igcn_cf_amd.ops import bpr_loss_terms
    rng = np.random.default_rng(d)
    nu, ni, B = 50, 80, 333
    Ut = (rng.standard_normal((nu, d)) * 0.3).astype(np.float32)
    It = (rng.standard_normal((ni, d)) * 0.3).astype(np.float32)
    users, pos, neg = rng.integers(0, nu, B), rng.integers(0, ni, B), rng.integers(0, ni, B)
    U = torch.nn.Parameter(_dev(Ut)); I = torch.nn.Parameter(_dev(It))
    terms = bpr_loss_terms(U, I, U, I, None, _dev(users), _dev(pos), _dev(neg))
    ue, pe, ne, l2 = O.bpr_forward_mf(Ut, It, users, pos, neg)
    bpr, _ = O.bpr_loss(ue, pe, ne, l2, 0.)
    assert abs(terms[0].item() - bpr) < 1e-5 and abs(terms[1].item() - float(l2.mean())) < 1e-4
    (terms[0] + 0.05 * terms[1]).backward()
    gu, gp, gn = O.bpr_grads(ue, pe, ne)
    GU = np.zeros((nu, d)); GI = np.zeros((ni, d))
    np.add.at(GU, users, gu + 0.05 * 2 / B * ue)
    np.add.at(GI, pos, gp + 0.05 * 2 / B * pe)
    np.add.at(GI, neg, gn + 0.05 * 2 / B * ne)
    np.testing.assert_allclose(U.grad.cpu().numpy(), GU, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(I.grad.cpu().numpy(), GI, rtol=1e-4, atol=1e-7)


# ---------------------------------------------------------------- sampler ------------------------------
def test_device_sampler_semantics(golden):
    """dataset.py:119-131: user has a non-empty train list, positive in it, negative not in it;
    users uniform over non-empty users."""
    from igcn_cf_amd.dataset import ProcessedDataset
    from igcn_cf_amd.trainer import DeviceSampler
    ds = ProcessedDataset({'name': 'ProcessedDataset', 'path': golden['path'], 'device': 'cuda'})
    sm = DeviceSampler(ds, 'cuda', seed=3)
    batches = [b.cpu().numpy() for b in sm.epoch_batches(100)]
    assert sum(len(b) for b in batches) == len(ds) and len(batches) == -(-len(ds) // 100)
    s = np.concatenate(batches + [b.cpu().numpy() for b in sm.epoch_batches(4000)])
    train = ds.train_data
    for u, p, n in s:
        assert train[u] and p in train[u] and n not in train[u] and 0 <= n < ds.n_items
    nonempty = [u for u in range(ds.n_users) if train[u]]
    cnt = np.bincount(s[:, 0], minlength=ds.n_users)[nonempty]
    exp = len(s) / len(nonempty)
    assert ((cnt - exp) ** 2 / exp).sum() < len(nonempty) + 6 * np.sqrt(2 * len(nonempty))   # chi-square
    assert not np.array_equal(batches[0], batches[1])


def test_device_sampler_batches_are_independent_across_roles():
    """with seeds s and s+1 the user draw of batch t lined up with the positive / negative draw of
    batch t+1.  With mixed per-batch seeds no pair of roles of consecutive batches is correlated."""
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.trainer import DeviceSampler
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'n_users': 5000, 'n_items': 4000, 'n_inter': 200000, 'device': 'cuda'})
    sm = DeviceSampler(ds, 'cuda', seed=2021)
    it = sm.epoch_batches(4096)
    a, b = next(it).cpu().numpy().astype(np.float64), next(it).cpu().numpy().astype(np.float64)
    rp, _ = ds.csr('train', sort=True)
    lens = np.diff(rp)

    def fractions(x):                                    # each role as a fraction in [0, 1): what the hash feeds
        u = x[:, 0].astype(np.int64)
        return np.stack([x[:, 0] / ds.n_users, (x[:, 1] % 97) / 97.0, x[:, 2] / ds.n_items], axis=1), lens[u]
    fa, _ = fractions(a)
    fb, _ = fractions(b)
    for shift in (0, 1, 3):                              # draw i against draw i, i^1, i^3 of the next batch
        idx = np.arange(len(fa)) ^ shift
        for ra in range(3):
            for rb in range(3):
                if ra == 1 or rb == 1:
                    continue                             # positives depend on the user's list, not a plain fraction
                c = np.corrcoef(fa[:, ra], fb[idx, rb])[0, 1]
                assert abs(c) < 0.08, (shift, ra, rb, c)


def test_eval_with_all_lists_empty_gives_nan_like_the_reference():
    """An evaluated split whose lists are all empty (trainer.py:109-138 then takes the mean of an empty
    selection: nan): igcn_hit_matrix gets no column array and reports no hits."""
    import torch
    from igcn_cf_amd import ops
    rec = torch.randint(0, 50, (8, 5), dtype=torch.int64, device='cuda')
    rp = torch.zeros(9, dtype=torch.int64, device='cuda')
    hit = ops.hit_matrix(rec, rp, torch.empty(0, dtype=torch.int32, device='cuda'))
    assert float(hit.abs().sum()) == 0.0
    from igcn_cf_amd.trainer import BasicTrainer

    class DS:
        n_users, n_items = 8, 50
    tr = BasicTrainer({'name': 'BasicTrainer', 'dataset': DS(), 'model': None, 'topks': [5], 'device': 'cuda',
                       'n_epochs': 0, 'test_batch_size': 7})
    m = tr._metrics_from_hits_device(hit, torch.zeros(8, dtype=torch.int64, device='cuda'))
    assert all(np.isnan(m[name][5]) for name in m)
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            m2 = tr.calculate_metrics([[] for _ in range(8)], rec.cpu().numpy())
    assert all(np.isnan(m2[name][5]) for name in m2)


def test_bpr_column_slices_sum_to_the_full_loss():
    """Embedding-column sharding: partial dots of the slices (igcn_bpr_dots_f32) summed by reduce_fn and
    finished (igcn_bpr_finish_f32) give the unsliced loss; each slice's gradient is the slice of the full one."""
    from igcn_cf_amd.ops import bpr_loss_terms
    rng = np.random.default_rng(9)
    n, nu, d, B = 120, 50, 64, 257
    rep = (rng.standard_normal((n, d)) * 0.3).astype(np.float32)
    emb = (rng.standard_normal((n, d)) * 0.3).astype(np.float32)
    users, pos, neg = rng.integers(0, nu, B), rng.integers(0, n - nu, B), rng.integers(0, n - nu, B)
    u_, p_, n_ = _dev(users), _dev(pos), _dev(neg)
    R, E = torch.nn.Parameter(_dev(rep)), torch.nn.Parameter(_dev(emb))
    full = bpr_loss_terms(R, R, E, E, None, u_, p_, n_, nu, nu)
    (full[0] + 0.1 * full[1]).backward()
    for P in (2, 4, 8):
        dl = d // P
        slices = [(torch.nn.Parameter(_dev(rep[:, r * dl:(r + 1) * dl].copy())), torch.nn.Parameter(_dev(emb[:, r * dl:(r + 1) * dl].copy())))
                  for r in range(P)]
        partial = []
        for Rs, Es in slices:                      # pass 1: every "rank" computes its partial dots
            bpr_loss_terms(Rs, Rs, Es, Es, None, u_, p_, n_, nu, nu, reduce_fn=lambda t: partial.append(t.clone()))
        total = torch.stack(partial).sum(0)
        for r, (Rs, Es) in enumerate(slices):      # pass 2: the all-reduce result is the sum of all partials
            terms = bpr_loss_terms(Rs, Rs, Es, Es, None, u_, p_, n_, nu, nu, reduce_fn=lambda t: t.copy_(total))
            assert abs(terms[0].item() - full[0].item()) < 1e-6 and abs(terms[1].item() - full[1].item()) < 1e-5
            (terms[0] + 0.1 * terms[1]).backward()
            np.testing.assert_allclose(Rs.grad.cpu().numpy(), R.grad.cpu().numpy()[:, r * dl:(r + 1) * dl], rtol=1e-4, atol=1e-8)
            np.testing.assert_allclose(Es.grad.cpu().numpy(), E.grad.cpu().numpy()[:, r * dl:(r + 1) * dl], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize('d', [64, 128])
def test_two_stage_sweep_order_and_early_exit_do_not_change_the_lists(d):
    """Round 3: the candidate sweep meets the items by descending norm and stops once no user of a wave can be reached
    by the rows still to come (|score| <= |u| |i|).  On tables whose row norms spread over orders of magnitude — where
    the exit skips most of the sweep and the last waves hand their remaining users to the fp32 sweep — with exclusion lists, banned items, duplicated rows, a zero user, users of very
    different scale and a ragged item count: the lists are those of the fp32 sweep, bit for bit, with the order and
    the exit switched on and off."""
    from igcn_cf_amd import _lib
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(11)
    n_users, n_items, k = 700, 41003, 20
    U = (rng.standard_normal((n_users, d)) * 0.1 * np.exp(rng.standard_normal((n_users, 1)))).astype(np.float32)
    I = (rng.standard_normal((n_items, d)) * 0.1 * np.exp(1.2 * rng.standard_normal((n_items, 1)))).astype(np.float32)
    I[300] = I[17]; I[40000] = I[17]
    I[5000:5040] = 0.0                                   # zero rows: tie at score 0 for every user
    U[9] = 0.0
    ex = [sorted(rng.choice(n_items, size=int(rng.integers(0, 60)), replace=False).tolist()) for _ in range(n_users)]
    top_norm = np.argsort(-(I * I).sum(1))[:200]
    for u in range(0, n_users, 3):                       # exclude some of the longest rows: the would-be winners
        ex[u] = sorted(set(ex[u]) | set(rng.choice(top_norm, size=30, replace=False).tolist()))
    rowptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum([len(x) for x in ex], out=rowptr[1:])
    col = np.array([i for x in ex for i in x], dtype=np.int32)
    bmask = np.zeros(n_items, dtype=np.uint8); bmask[rng.choice(n_items, size=n_items // 9, replace=False)] = 1
    users = rng.permutation(n_users).astype(np.int64)
    kw = dict(user_ids=_dev(users), excl_rowptr=_dev(rowptr), excl_col=_dev(col), banned=_dev(bmask))
    ref = score_topk(_dev(U), _dev(I), k, mode='exact', **kw)
    try:
        # (sweep order, early exit, user planes, candidates kept beyond k, stragglers give up): library defaults = None
        for order, ex_it, planes, extra, give_up in ((None, None, None, None, None), (None, 0, None, None, None), (0, None, None, None, None),
                                                    (None, None, 2, None, None), (None, None, None, 1, None), (None, None, None, 8, 0),
                                                    (None, None, 2, 7, None), (None, None, None, None, 0)):
            _lib.set_tuning('topk_fast_order', order)
            _lib.set_tuning('topk_fast_exit', ex_it)
            _lib.set_tuning('topk_fast_mode', planes)
            _lib.set_tuning('topk_fast_extra', extra)
            _lib.set_tuning('topk_fast_give_up', give_up)
            got = score_topk(_dev(U), _dev(I), k, mode='fast', **kw)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (order, ex_it, planes, extra, give_up)
            got = score_topk(_dev(U), _dev(I), k, mode='fast')          # no masks
            ref0 = score_topk(_dev(U), _dev(I), k, mode='exact')
            assert torch.equal(got[0], ref0[0]) and torch.equal(got[1], ref0[1]), (order, ex_it, planes, extra, give_up)
    finally:
        _lib.set_tuning('topk_fast_order', None)
        _lib.set_tuning('topk_fast_exit', None)
        _lib.set_tuning('topk_fast_mode', None)
        _lib.set_tuning('topk_fast_extra', None)
        _lib.set_tuning('topk_fast_give_up', None)


@pytest.mark.parametrize('d', [64, 128])
def test_two_stage_call_does_not_rely_on_what_its_workspace_held_and_survives_unzeroed_bins(d):
    """Two properties of igcn_score_topk_fast_f32's workspace.  (1) The call clears what it needs itself: with the workspace filled
    with 0x5A / 0xFF bytes before the call the lists are the fp32 sweep's, bit for bit.  (2) The order build's counting bins must
    arrive zeroed — round 4's captured memset node once left them unzeroed, a GPU fault then, a silently wrong permutation behind
    round 5's bounds guard.  With the test-only knob "topk_fast_poison" the call leaves the bins as the workspace held
    them (garbage): the build notices that its counts no longer add up to n_items, every workgroup falls back to the identity
    order, the tile bounds stop any early exit — the SAME lists come back (a slower sweep), trained-like norms, masks and all."""
    from igcn_cf_amd import _lib, ops
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(23)
    n_users, n_items, k = 900, 30011, 20
    U = (rng.standard_normal((n_users, d)) * 0.1 * np.exp(rng.standard_normal((n_users, 1)))).astype(np.float32)
    I = (rng.standard_normal((n_items, d)) * 0.1 * np.exp(1.2 * rng.standard_normal((n_items, 1)))).astype(np.float32)
    ex = [sorted(rng.choice(n_items, size=int(rng.integers(0, 40)), replace=False).tolist()) for _ in range(n_users)]
    rowptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum([len(x) for x in ex], out=rowptr[1:])
    col = np.array([i for x in ex for i in x], dtype=np.int32)
    kw = dict(excl_rowptr=_dev(rowptr), excl_col=_dev(col))
    ref = score_topk(_dev(U), _dev(I), k, mode='exact', **kw)
    try:
        for fill, poison in ((0x5A, None), (0xFF, None), (0x5A, 1), (0x01, 1), (0xFF, 1)):
            ops.FAST_WORKSPACE_FILL = fill
            _lib.set_tuning('topk_fast_poison', poison)
            got = score_topk(_dev(U), _dev(I), k, mode='fast', **kw)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (fill, poison)
    finally:
        ops.FAST_WORKSPACE_FILL = None
        _lib.set_tuning('topk_fast_poison', None)
    got = score_topk(_dev(U), _dev(I), k, mode='fast', **kw)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


def test_fused_eval_metrics_match_the_reference_formulas(golden):
    """igcn_eval_metrics (one pass over the recommended lists) against calculate_metrics' numpy restatement — which the
    reference-produced fixtures pin bit for bit (test_hit_matrix_and_metrics_golden) — on the golden recommendations and on
    random lists with empty users, several cut-offs, cut-off == list width; and the all-empty split (nan)."""
    from igcn_cf_amd import ops
    from igcn_cf_amd.trainer import BasicTrainer
    rng = np.random.default_rng(3)

    def check(lists, rec, topks, n_items):
        class DS:
            n_users = len(lists)
        DS.n_items = n_items
        tr = BasicTrainer({'name': 'BasicTrainer', 'dataset': DS(), 'model': None, 'topks': topks, 'device': 'cuda',
                           'n_epochs': 0, 'test_batch_size': 7})
        with np.errstate(all='ignore'):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                want = tr.calculate_metrics(lists, rec)
        rowptr = np.zeros(len(lists) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in lists], out=rowptr[1:])
        col = np.array([i for x in lists for i in sorted(x)], dtype=np.int32)
        got = tr._metrics_device(_dev(rec.astype(np.int64)), _dev(rowptr), _dev(col) if col.size else None)
        for name in want:
            for k in topks:
                a, b = float(got[name][k]), float(want[name][k])
                assert (np.isnan(a) and np.isnan(b)) or abs(a - b) < 1e-6, (name, k, a, b)

    lists = _lists(golden)
    topks = [int(k) for k in golden['eval_topks']]
    for tag, stage in (('train', 'train'), ('val', 'val'), ('test', 'test'), ('testban', 'test')):
        check(lists[stage], np.asarray(golden['eval_%s_rec' % tag]), topks, int(golden['n_items']))
    n_users, n_items = 5000, 700
    lists = [sorted(rng.choice(n_items, size=int(rng.integers(0, 40)) if u % 7 else 0, replace=False).tolist()) for u in range(n_users)]
    rec = np.stack([rng.permutation(n_items)[:50] for _ in range(n_users)])
    check(lists, rec, [1, 5, 20, 50], n_items)
    check(lists, rec[:, :20], [20], n_items)
    check([[] for _ in range(300)], rec[:300, :10], [5, 10], n_items)


@pytest.mark.parametrize('d', [64, 128])
def test_two_stage_evaluation_at_full_amazon_size_is_the_fp32_sweep(d):
    """BASELINE's evaluation shape (109 730 users x 96 421 items, k = 20, train + val lists excluded) at both widths the
    two-stage path takes: ids and scores of the fp32 sweep bit for bit; the lists hold no excluded item, are ordered, and
    re-scored in float64 they are the float64 top-20 up to fp32 rounding (a sample of users)."""
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.ops import score_topk
    from igcn_cf_amd.trainer import _csr_to_device, _merge_sorted_csr
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
    g = torch.Generator(device='cuda').manual_seed(d)
    U = torch.randn(ds.n_users, d, device='cuda', generator=g) * 0.1
    I = torch.randn(ds.n_items, d, device='cuda', generator=g) * 0.1
    excl = _merge_sorted_csr(ds.csr('train'), ds.csr('val'))
    rp, cl = _csr_to_device(excl[0], excl[1], 'cuda')
    a = score_topk(U, I, 20, excl_rowptr=rp, excl_col=cl, mode='fast')
    flagged = score_topk.last_flagged
    b = score_topk(U, I, 20, excl_rowptr=rp, excl_col=cl, mode='exact')
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert flagged < ds.n_users // 50                     # the fall-back stays the exception (43 / 173 users on this data)
    assert bool((a[1][:, :-1] >= a[1][:, 1:]).all())
    sample = torch.arange(0, ds.n_users, 997, device='cuda')
    s64 = U[sample].double() @ I.double().T
    rows = torch.repeat_interleave(torch.arange(ds.n_users, device='cuda'), rp[1:] - rp[:-1])
    keep = torch.isin(rows, sample)
    pos = torch.searchsorted(sample, rows[keep])
    s64[pos, cl[keep].long()] = -float('inf')
    assert bool((torch.gather(s64, 1, a[0][sample]) > -float('inf')).all())          # nothing excluded was recommended
    top = torch.topk(s64, 20, dim=1).values
    got = torch.gather(s64, 1, a[0][sample])
    assert float((top - got).abs().max()) < 1e-6


@pytest.mark.parametrize('d', [64, 128])
def test_two_stage_stragglers_hand_their_users_to_the_fp32_sweep(d):
    """The state of a trained recommender: every user likes the long (popular) rows, so nearly every wave of the candidate
    sweep leaves after a few tiles — except the waves that hold one of a few users whose scores are all small, which would
    crawl through the whole table alone.  Those give up once three quarters of the waves are gone and their users are
    re-done by the fp32 sweep: the lists are the fp32 sweep's all the same, with and without giving up, with masks."""
    from igcn_cf_amd import _lib
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(17)
    n_users, n_items, k = 6400, 40000, 20
    pop = rng.standard_normal(d).astype(np.float32)
    pop /= np.linalg.norm(pop)
    scale = np.exp(1.0 * rng.standard_normal((n_items, 1))).astype(np.float32)
    I = (0.05 * rng.standard_normal((n_items, d)) + 0.3 * scale * pop[None, :]).astype(np.float32)
    U = (0.05 * rng.standard_normal((n_users, d)) + 1.0 * pop[None, :]).astype(np.float32)
    odd = rng.choice(n_users, size=25, replace=False)
    U[odd] = (0.3 * rng.standard_normal((25, d))).astype(np.float32)                       # no taste for the popular rows
    ex = [sorted(rng.choice(n_items, size=int(rng.integers(0, 30)), replace=False).tolist()) for _ in range(n_users)]
    rowptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum([len(x) for x in ex], out=rowptr[1:])
    col = np.array([i for x in ex for i in x], dtype=np.int32)
    kw = dict(excl_rowptr=_dev(rowptr), excl_col=_dev(col))
    Ud, Id = _dev(U), _dev(I)
    ref = score_topk(Ud, Id, k, mode='exact', **kw)
    try:
        # (early: exit checks every 6 tiles up to tile 48 and give-up from tile 12, the default; 0: every 24 tiles / from tile 48; 3: every 3)
        for give_up, narrow, early in ((None, None, None), (0, None, None), (None, 0, None), (None, 0, 0), (None, 0, 3), (None, None, 0)):
            _lib.set_tuning('topk_fast_give_up', give_up)
            _lib.set_tuning('topk_fast_early_checks', early)
            _lib.set_tuning('topk_fast_narrow', narrow)            # 0: 64-user wave-groups although the batch is small
            got = score_topk(Ud, Id, k, mode='fast', **kw)
            flagged = score_topk.last_flagged
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), give_up
            if give_up is None:
                assert 1 <= flagged <= 400, flagged               # the stragglers' users, not whole waves' worth of the batch
    finally:
        _lib.set_tuning('topk_fast_give_up', None)
        _lib.set_tuning('topk_fast_narrow', None)
        _lib.set_tuning('topk_fast_early_checks', None)


@pytest.mark.parametrize('d', [64, 128])
def test_two_stage_fall_back_planned_on_the_device(d):
    """ABI v7: igcn_score_topk_fast_f32 finishes its first 256 flagged users itself — the bounded fp32 sweep is launched behind the
    re-scoring kernel, planned for 256 users and run for as many as the device-side count holds (its batch is the flagged list);
    the caller re-does only what lies beyond.  No flagged user, a few, exactly around 256 and thousands (integer tables tie in
    droves), with masks, banned items and a permuted user subset: the lists are the fp32 sweep's, and the same as with the
    whole fall-back done from the host (the ABI v6 split)."""
    from igcn_cf_amd import _lib, ops
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(43)
    n_users, n_items, k = 3000, 9000, 20
    ex = [np.sort(rng.choice(n_items, size=int(rng.integers(0, 30)), replace=False)) for _ in range(n_users)]
    rowptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum([len(x) for x in ex], out=rowptr[1:])
    bmask = np.zeros(n_items, dtype=np.uint8)
    bmask[rng.choice(n_items, size=n_items // 9, replace=False)] = 1
    kw = dict(excl_rowptr=_dev(rowptr), excl_col=_dev(np.concatenate(ex).astype(np.int32)), banned=_dev(bmask))
    sub = _dev(rng.permutation(n_users)[:1700].astype(np.int64))
    gauss_u, gauss_i = (rng.standard_normal((n_users, d)) * 0.1).astype(np.float32), (rng.standard_normal((n_items, d)) * 0.1).astype(np.float32)
    # rows in {-1, 0, 1} on eight columns, zero elsewhere: scores are small integers and tie in droves — such a user's candidate
    # list ends inside a run of equal scores and cannot be proven complete
    def coarse(n):
        t = np.zeros((n, d), dtype=np.float32)
        t[:, :8] = rng.integers(-1, 2, size=(n, 8))
        return t
    coarse_i = coarse(n_items)
    mixed_u = gauss_u.copy()
    mixed_u[:, 8:] = 0                                                       # Gaussian on the items' eight live columns: no ties ...
    mixed_u[:270] = coarse(270)                                              # ... except for these 270 users
    cases = {'gaussian (a handful flagged)': (gauss_u, gauss_i),
             'coarse items, 270 coarse users (around the 256 the call finishes itself)': (mixed_u, coarse_i),
             'all coarse (every user flagged)': (coarse(n_users), coarse_i)}
    seen = []
    try:
        for name, (U, I) in cases.items():
            Ud, Id = _dev(U), _dev(I)
            for users in (None, sub):
                ref = score_topk(Ud, Id, k, user_ids=users, mode='exact', **kw)
                # (inside the call the flagged users first meet the streaming filter — every (user, item) pair scored once, the pairs that
                # reach the user's bound kept — and only those whose lists overflow it, or whose bound is -inf, the bounded sweep;
                # "topk_fast_filter" 0: the bounded sweep for all of them, as until late round 4)
                for inside, filt in ((True, None), (True, 0), (False, None)):
                    ops.set_fast_fallback(inside)
                    _lib.set_tuning('topk_fast_filter', filt)
                    got = score_topk(Ud, Id, k, user_ids=users, mode='fast', **kw)
                    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (name, inside, filt)
                seen.append(score_topk.last_flagged)
    finally:
        ops.set_fast_fallback(True)
        _lib.set_tuning('topk_fast_filter', None)
    assert min(seen) <= 64 and max(seen) > 1000 and any(150 < n < 400 for n in seen), seen       # below, around and far above 256


@pytest.mark.parametrize('d', [64, 128])
def test_two_stage_candidate_count_at_the_upper_end_of_k(d):
    """k = 57 ... 60 on Gaussian tables: the candidate sweep keeps k + 6 candidates while they fit the 64 lanes of the
    re-scoring wave (k <= 58) and 64 - k beyond (5 at k = 59, 4 at k = 60 — more users then fail the completeness check and
    take the fp32 sweep); k = 61 is refused by the two-stage path and taken by the fp32 sweep under mode='auto'.  The lists
    are the fp32 sweep's at every k."""
    from igcn_cf_amd import _lib
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(41)
    n_users, n_items = 700, 12000
    U, I = _dev((rng.standard_normal((n_users, d)) * 0.1).astype(np.float32)), _dev((rng.standard_normal((n_items, d)) * 0.1).astype(np.float32))
    flagged = {}
    for k in (57, 58, 59, 60):
        a, b = score_topk(U, I, k, mode='fast'), score_topk(U, I, k, mode='exact')
        flagged[k] = score_topk.last_flagged
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), k
    assert flagged[58] <= n_users // 4, flagged                    # six spare candidates: few users fail the check
    with pytest.raises(_lib.IgcnError):
        score_topk(U, I, 61, mode='fast')
    a, b = score_topk(U, I, 61, mode='auto'), score_topk(U, I, 61, mode='exact')
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_bounded_sweep_with_a_bound_that_is_too_high_falls_back_to_the_plain_sweep():
    """igcn_score_topk_bounded_f32 only looks at items that reach the caller's bound.  A bound above the true k-th best
    score (by an ulp, or plainly invalid) must cost time, not correctness: ops.score_topk redoes the users whose lists came
    out short with the plain fp32 sweep.  Exact bounds, bounds an ulp high, +inf, and valid ones, mixed in one batch."""
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(31)
    n_users, n_items, d, k = 200, 5000, 64, 20
    U, I = _dev((rng.standard_normal((n_users, d)) * 0.1).astype(np.float32)), _dev((rng.standard_normal((n_items, d)) * 0.1).astype(np.float32))
    ex = [np.sort(rng.choice(n_items, size=int(rng.integers(0, 40)), replace=False)) for _ in range(n_users)]
    rowptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum([len(x) for x in ex], out=rowptr[1:])
    kw = dict(excl_rowptr=_dev(rowptr), excl_col=_dev(np.concatenate(ex).astype(np.int32)))
    ref_idx, ref_val = score_topk(U, I, k, mode='exact', **kw)
    kth = ref_val[:, k - 1].clone()
    bound = kth.clone()                                                        # exact: the k-th best itself (>= keeps it)
    bound[0::4] = torch.nextafter(kth[0::4], torch.full_like(kth[0::4], float('inf')))      # an ulp too high
    bound[1::4] = float('inf')                                                               # nothing reaches it
    bound[2::4] = kth[2::4] - 0.01                                                           # valid, loose
    got_idx, got_val = score_topk(U, I, k, mode='exact', lower_bound=bound.contiguous(), **kw)
    assert torch.equal(got_idx, ref_idx) and torch.equal(got_val, ref_val)
    assert int((got_idx < 0).sum()) == 0


def test_two_stage_second_whole_sweep_of_a_wave_counts_its_own_leavers():
    """Several whole sweeps per wave (more 64-user groups than wave slots: above 131 072 users per call on the MI355X; here
    the slots are tuned down): the give-up rule compares the waves that have left THIS job with three quarters of the grid.
    Counted per call instead (round 3), the leavers of job 0 pushed every wave of job 1 that was alive at the first check
    to hand its users over.  On a trained-like table (log-normal row scales: most waves leave within a few checks) the users
    handed to the fp32 sweep must stay a few stragglers whatever the number of jobs; the lists are the fp32 sweep's."""
    from igcn_cf_amd import _lib
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(29)
    d, n_users, n_items, k = 64, 64 * 96, 40000, 20
    pop = rng.standard_normal(d).astype(np.float32)
    pop /= np.linalg.norm(pop)
    scale = np.exp(1.0 * rng.standard_normal((n_items, 1))).astype(np.float32)
    I = (0.05 * rng.standard_normal((n_items, d)) + 0.3 * scale * pop[None, :]).astype(np.float32)
    U = (0.05 * rng.standard_normal((n_users, d)) + 1.0 * pop[None, :]).astype(np.float32)
    odd = rng.choice(n_users, size=12, replace=False)
    U[odd] = (0.3 * rng.standard_normal((12, d))).astype(np.float32)                       # the stragglers: no taste for the popular rows
    Ud, Id = _dev(U), _dev(I)
    ref = score_topk(Ud, Id, k, mode='exact')
    flagged = {}
    try:
        _lib.set_tuning('topk_fast_narrow', 0)                                              # 64-user wave-groups: 96 of them
        for slots in (96, 48, 32, 40):          # 1, 2, 3 whole sweeps per wave; 40: two whole sweeps + a rest cut into pieces
            _lib.set_tuning('topk_slots', slots)
            got = score_topk(Ud, Id, k, mode='fast')
            flagged[slots] = score_topk.last_flagged
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), slots
    finally:
        _lib.set_tuning('topk_slots', None)
        _lib.set_tuning('topk_fast_narrow', None)
    # a handed-over wave carries up to 64 users; 12 stragglers in at most 12 waves.  With the per-call count every wave of the
    # later jobs that was alive at tile 48 handed over its 64 users: thousands.
    assert all(v <= 12 * 64 + 64 for v in flagged.values()), flagged
    assert flagged[48] <= flagged[96] + 6 * 64 and flagged[32] <= flagged[96] + 6 * 64, flagged


def test_two_stage_call_replays_from_a_captured_hip_graph():
    """igcn_score_topk_fast_f32 makes no host read, no allocation and no runtime memset, and no kernel of it carries a private segment (include/igcn_hip.h,
    ABI v8): the whole evaluation — row statistics, order build, row sorts, candidate sweep, re-scoring, the flagged users' filter and
    bounded sweep — can be captured into ONE HIP graph by PyTorch's documented recipe (warm-up on a side stream, capture, replay on the
    current stream) and replayed on new table contents; so can the fp32 sweep (igcn_score_topk_f32).  Run in a FRESH child process
    (tests/capture_child.py), where the replaying stream has run nothing eagerly — the sequence that faulted the GPU in round 4 (the
    call's captured hipMemsetAsync, a memset node, was not ordered against the kernels behind it: csrc/common.h, zero_async).  d = 64 and 128, three refills each (flat norms: warm-up pass; spread norms: early exits): the lists
    are the fp32 sweep's every time and flagged[0] stays within what the call finishes itself.  The child also checks that
    igcn_csr_transpose (rocPRIM's sort, the one kernel family with scratch) REFUSES a capturing stream with IGCN_E_CAPTURE."""
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'capture_child.py')
    p = subprocess.run([sys.executable, child], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, (p.returncode, out[-500:], p.stderr.decode()[-2000:])
    assert 'ok 6' in out and 'ok refused' in out, out


def test_two_stage_order_build_on_degenerate_norm_distributions():
    """The sweep order is a counting sort of the item norms on 65 536 bins (csrc/topk_order.hip).  Tables whose norms fall into ONE
    bin (unit-length rows: every lane of every wave goes to the same counter), into two, into every binade from 2^-60 to 2^60, an
    all-zero table, and tables smaller than a wave / a tile: the lists are the fp32 sweep's, with exclusion lists and a user subset."""
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(61)
    d, n_users, k = 64, 900, 10

    def unit(n):
        x = rng.standard_normal((n, d)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    U = _dev((rng.standard_normal((n_users, d)) * 0.1).astype(np.float32))
    two = unit(30000)
    two[::3] *= 2.0
    wide = unit(30000) * np.exp2(rng.integers(-60, 61, size=(30000, 1))).astype(np.float32)
    tables = {'one bin': unit(30000), 'two bins': two, 'every binade': wide, 'all zero': np.zeros((5000, d), dtype=np.float32),
              '40 items': unit(40), '11 items': unit(11) * 3.0, '33 items, equal rows': np.tile(unit(1), (33, 1))}
    for name, I in tables.items():
        n_items = I.shape[0]
        ex = [np.sort(rng.choice(n_items, size=int(rng.integers(0, min(20, n_items - k))), replace=False)) for _ in range(n_users)]
        rowptr = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum([len(x) for x in ex], out=rowptr[1:])
        masks = dict(excl_rowptr=_dev(rowptr), excl_col=_dev(np.concatenate(ex).astype(np.int32)))
        sub = _dev(rng.permutation(n_users)[:500].astype(np.int64))
        Id = _dev(I)
        for kw in ({}, masks, dict(user_ids=sub, **masks)):
            ref = score_topk(U, Id, k, mode='exact', **kw)
            got = score_topk(U, Id, k, mode='fast', **kw)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (name, sorted(kw))


def test_two_stage_warm_up_pass_bounds_do_not_change_the_lists():
    """Whole candidate sweeps start with a warm-up pass ("topk_fast_warm" tiles): the kc-th largest of a user's 32 slot maxima over
    those tiles becomes the threshold the sweep proper starts from.  The bound must hold with exclusion lists that take away the
    user's best items of exactly those tiles, with banned items, with users that have fewer unmasked items than slots (bound = -inf),
    with scores that tie in droves, on Gaussian and trained-like tables (where waves leave early: a list the bound has not filled
    keeps its wave in the sweep), for one and several whole sweeps per wave and every warm-up length from 1 tile to all but one:
    the lists are the fp32 sweep's, bit for bit."""
    from igcn_cf_amd import _lib
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(53)
    d, n_users, n_items = 64, 64 * 24, 6000
    pop = rng.standard_normal(d).astype(np.float32)
    pop /= np.linalg.norm(pop)
    scale = np.exp(1.0 * rng.standard_normal((n_items, 1))).astype(np.float32)
    tables = {'gaussian': ((rng.standard_normal((n_users, d)) * 0.1).astype(np.float32), (rng.standard_normal((n_items, d)) * 0.1).astype(np.float32)),
              'trained-like': ((0.05 * rng.standard_normal((n_users, d)) + 1.0 * pop[None, :]).astype(np.float32),
                               (0.05 * rng.standard_normal((n_items, d)) + 0.3 * scale * pop[None, :]).astype(np.float32))}
    # (the pass is taken only where the rows at its end are still half as long as the first: not on the table above, but on this one —
    # whose waves still leave before the end of the table)
    mild = np.exp(0.12 * rng.standard_normal((n_items, 1))).astype(np.float32)
    tables['mildly trained-like'] = ((0.02 * rng.standard_normal((n_users, d)) + 1.0 * pop[None, :]).astype(np.float32),
                                     (0.02 * rng.standard_normal((n_items, d)) + 0.3 * mild * pop[None, :]).astype(np.float32))
    coarse_u, coarse_i = np.zeros((n_users, d), dtype=np.float32), np.zeros((n_items, d), dtype=np.float32)
    coarse_u[:, :8] = rng.integers(-1, 2, size=(n_users, 8))
    coarse_i[:, :8] = rng.integers(-1, 2, size=(n_items, 8))
    coarse_i *= (1.0 + 0.001 * rng.random((n_items, 1))).astype(np.float32)        # (distinct norms: the sweep order is not the id order)
    tables['small integers (ties)'] = (coarse_u, coarse_i)
    bmask = np.zeros(n_items, dtype=np.uint8)
    bmask[rng.choice(n_items, size=n_items // 7, replace=False)] = 1
    try:
        _lib.set_tuning('topk_fast_narrow', 0)
        for name, (U, I) in tables.items():
            Ud, Id = _dev(U), _dev(I)
            # every user's exclusion list: its 0 ... 40 best items (so the warm-up tiles' best scores are masked ones), some random
            # ones; a few users keep fewer than k items at all
            best = np.argsort(-(U[:, :] @ I.T), axis=1)[:, :40]
            ex = [np.unique(np.concatenate([best[u, :int(rng.integers(0, 41))], rng.choice(n_items, size=int(rng.integers(0, 20)), replace=False)])) for u in range(n_users)]
            for u in rng.choice(n_users, size=6, replace=False):
                ex[u] = np.setdiff1d(np.arange(n_items), rng.choice(n_items, size=int(rng.integers(3, 30)), replace=False))
            rowptr = np.zeros(n_users + 1, dtype=np.int64)
            np.cumsum([len(x) for x in ex], out=rowptr[1:])
            masks = dict(excl_rowptr=_dev(rowptr), excl_col=_dev(np.concatenate(ex).astype(np.int32)), banned=_dev(bmask))
            for kw in ({}, masks):
                for k in (20, 1, 26):                                                   # k + extra = 26, 7, 32 (the most the pass takes)
                    ref = score_topk(Ud, Id, k, mode='exact', **kw)
                    for slots, warm in ((24, 0), (24, 1), (24, 2), (24, 63), (24, 128), (24, 186), (12, 64), (8, 30), (16, 64)):
                        _lib.set_tuning('topk_slots', slots)                            # 24 groups: 1, 2, 3 whole sweeps per wave; 16: one + pieces
                        _lib.set_tuning('topk_fast_warm', warm)
                        got = score_topk(Ud, Id, k, mode='fast', **kw)
                        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (name, bool(kw), k, slots, warm)
    finally:
        _lib.set_tuning('topk_slots', None)
        _lib.set_tuning('topk_fast_warm', None)
        _lib.set_tuning('topk_fast_narrow', None)


def test_two_stage_exclusion_lists_sorted_on_the_device_hold_every_users_best_items():
    """The exclusion lists reach the candidate sweep as sweep POSITIONS, sorted row by row on the device (csrc/topk_order.hip:
    half-wave rank sorts up to 32 entries, wave rank sorts up to 256, a workgroup's bitonic network in LDS up to 8 192 and in
    place in HBM beyond; which kernel takes a row is decided on the device, no host read).  Here user u excludes exactly its
    OWN L best items, L at and around every class boundary: one entry that is lost, duplicated or out of order puts an
    excluded item into the list or drops a rightful one.  Lists equal to the fp32 sweep's and to the float64 ranking."""
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(47)
    n_items, d, k = 45000, 64, 20
    lengths = [0, 1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1000, 4095, 8191, 8192, 8193, 20000, n_items - k, 1023, 1024, 1025, 500, 700]
    lengths = lengths + [int(x) for x in rng.integers(1, 300, size=45)]
    n_users = len(lengths)
    U = (rng.standard_normal((n_users, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((n_items, d)) * 0.1 * np.exp(0.3 * rng.standard_normal((n_items, 1)))).astype(np.float32)
    s64 = U.astype(np.float64) @ I.astype(np.float64).T
    order = np.argsort(-s64, axis=1, kind='stable')
    ex = [np.sort(order[u, :L]) for u, L in enumerate(lengths)]                  # ascending ids, as the API wants them
    rowptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(lengths, out=rowptr[1:])
    kw = dict(excl_rowptr=_dev(rowptr), excl_col=_dev(np.concatenate(ex).astype(np.int32)))
    # a permuted user subset as well: rows are looked up by user id, not by batch position
    # ... and a subset with repeats: only the rows of the call's users are sorted (flagged by a kernel over the batch), each once
    part = rng.choice(n_users, size=n_users // 2, replace=False)
    part = np.concatenate([part, part[:7], [17, 17, 18]]).astype(np.int64)          # (17, 18: the 20 000-entry and the all-but-k rows)
    for users in (None, _dev(rng.permutation(n_users).astype(np.int64)), _dev(part)):
        a = score_topk(_dev(U), _dev(I), k, user_ids=users, mode='fast', **kw)
        b = score_topk(_dev(U), _dev(I), k, user_ids=users, mode='exact', **kw)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        got = a[0].cpu().numpy()
        ids = np.arange(n_users) if users is None else users.cpu().numpy()
        for row, u in enumerate(ids):
            L = lengths[u]
            want = order[u, L:L + k]
            gap = np.abs(np.diff(s64[u, order[u, L:L + k + 1]])).min()           # a float64 near-tie may swap neighbours in fp32
            if gap > 1e-6:
                np.testing.assert_array_equal(got[row], want, err_msg='user %d, %d exclusions' % (u, L))
            else:
                assert set(got[row]) <= set(order[u, L:L + k + 2].tolist())


def test_two_stage_exclusion_lists_of_every_length_class():
    """The candidate sweep walks each user's exclusion list in SWEEP positions, sorted row by row on the device (a segmented
    radix sort): empty lists, short ones, lists of thousands of entries and one that leaves exactly k items.  The lists are
    the fp32 sweep's."""
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(23)
    n_users, n_items, d, k = 300, 45000, 64, 20
    U = (rng.standard_normal((n_users, d)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((n_items, d)) * 0.1 * np.exp(0.5 * rng.standard_normal((n_items, 1)))).astype(np.float32)
    sizes = rng.integers(0, 64, size=n_users)
    sizes[::7] = rng.integers(65, 2049, size=len(sizes[::7]))
    sizes[3], sizes[10], sizes[50], sizes[100], sizes[200] = 2049, 16384, 16385, 30000, n_items - k
    sizes[5], sizes[6], sizes[77] = 0, 64, 2048
    ex = [np.sort(rng.choice(n_items, size=int(m), replace=False)) for m in sizes]
    rowptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(sizes, out=rowptr[1:])
    col = np.concatenate(ex).astype(np.int32)
    kw = dict(excl_rowptr=_dev(rowptr), excl_col=_dev(col))
    a = score_topk(_dev(U), _dev(I), k, mode='fast', **kw)
    b = score_topk(_dev(U), _dev(I), k, mode='exact', **kw)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    got = a[0].cpu().numpy()
    for u in (3, 10, 50, 100, 200, 6, 77):
        assert not np.isin(got[u], ex[u]).any(), u
    assert set(got[200].tolist()) == set(np.setdiff1d(np.arange(n_items), ex[200]).tolist())     # exactly the k items left


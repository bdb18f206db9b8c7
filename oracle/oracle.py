"""CPU oracle for the INMO / LightGCN propagation + scoring path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``igcn_cf_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

What this is
------------
A plain numpy restatement of the reference algorithm, written from reading the
reference source as text.  Every function cites the reference file:line it
follows (paths are relative to the reference tree, WuYunfan/igcn_cf @ v1).

How it is pinned
----------------
* ``utils.py``, ``dataset.py`` and ``trainer.py`` of the reference import
  unmodified in the build container.  ``oracle/gen_golden.py`` runs them and
  commits their outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
  checks this restatement against those vectors (adjacency build, template
  ranking, metrics, eval masking / top-k, BPR + aux loss arithmetic, dataset
  reader, auxiliary re-indexing).
* ``model.py`` of the reference (round 3): its only missing import is DGL (``README.md:16``
  "DGL >= 0.8", no lock file, not vendored, not in the image).  No stand-in for DGL
  exists anywhere in this repo.  ``gen_golden.py:model_fixtures`` lets the ``import dgl``
  STATEMENT pass by registering an inert module object — no attributes; touching one
  raises and is logged; the generator asserts the log is empty — and records what the
  reference's DGL-free functions return: ``LightGCN.generate_graph``, ``IGCN.generate_feat``
  (ratios 1 / 0.5 / 0.3, 'degree' / 'sort', and ``is_updating=True`` after the dropui /
  dropit live-model update), ``update_feat_mat`` / ``feat_mat_anneal``,
  ``NGCF.dropout_sp_mat`` (eval identity; train mask and 1/(1-p) scaling),
  ``bpr_forward`` / ``predict`` of MF / LightGCN / IGCN with ``get_rep`` replaced by a
  recorded tensor (which rows the L2 term reads), ``IGCN.save`` / ``load``.
  ``tests/test_model_golden.py`` checks ``lightgcn_norm_adj``, ``igcn_generate_feat``,
  ``igcn_feat_values``, ``dropout_keep_scale``, ``bpr_forward_*`` and ``predict`` below
  against those vectors (indices / maps / row sums / A_hat values bit-exact).
* BASELINE config 1 (MF + BPRTrainer + eval) is recorded END TO END from the reference's own
  classes (``e2e_mf_*`` keys of the same fixtures): batches, epoch losses, trained tables,
  recommended ids, metrics.
* STILL UNPINNED — the two gspmm callers only: ``lightgcn_get_rep`` (``model.py:96-106``)
  and the product inside ``igcn_get_rep`` (``model.py:423-446``).  The sparse product is
  DGL's ``gspmm(g, 'mul', 'sum', X, w)`` on ``dgl.graph((column, row))`` (call sites
  ``model.py:102``, ``:430``, ``:442``).  Its published semantics are
  out[dst] = sum over edges e=(src->dst) of X[src] * w[e],  i.e.  Y = M @ X  for the COO
  matrix (row=dst, col=src, val=w); ``spmm_coo`` below restates exactly that, and the
  layer mean around it (``torch.stack(...).mean(0)``) is plain torch.  PARITY UNPINNED at
  this one boundary: the reference holds no test or golden vector for it and DGL cannot
  be run here.  Tolerance for the fp32 product is 1e-4 relative (BASELINE.json).
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# graph construction
# --------------------------------------------------------------------------
def coo_sum_duplicates(row, col, val, shape):
    """Sorted row-major COO with duplicate (row, col) entries summed.

    This is what ``scipy.sparse.coo_matrix(...).tocsr()`` followed by
    ``tocoo()`` + ``torch.sparse...coalesce()`` produce (utils.py:32-38,
    utils.py:46-48).  Returns (row, col, val) with int64 indices.
    """
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val)
    if row.size == 0:
        return row, col, val.astype(F32)
    key = row * np.int64(shape[1]) + col
    order = np.argsort(key, kind='stable')
    key = key[order]
    val = val[order]
    first = np.ones(key.shape, dtype=bool)
    first[1:] = key[1:] != key[:-1]
    starts = np.flatnonzero(first)
    # scipy sums duplicates in the matrix dtype (float32 here)
    out_val = np.add.reduceat(val.astype(F32), starts).astype(F32)
    ukey = key[starts]
    return ukey // shape[1], ukey % shape[1], out_val


def generate_adj(train_array, n_users, n_items):
    """A = [[0, R], [R^T, 0]] with duplicates summed.  utils.py:41-49."""
    ta = np.asarray(train_array, dtype=np.int64).reshape(-1, 2)
    users, items = ta[:, 0], ta[:, 1]
    row = np.concatenate([users, items + n_users])
    col = np.concatenate([items + n_users, users])
    n = n_users + n_items
    return coo_sum_duplicates(row, col, np.ones(row.shape, dtype=F32), (n, n))


def lightgcn_norm_adj(train_array, n_users, n_items):
    """A_hat = D^-1/2 A D^-1/2 with deg = max(1, rowsum).  model.py:85-94.

    All arithmetic in float32 in the reference's order:
    ``d_mat.dot(adj_mat).dot(d_mat)`` = fl(fl(d[r]*a)*d[c]).
    """
    row, col, val = generate_adj(train_array, n_users, n_items)
    n = n_users + n_items
    degree = np.zeros(n, dtype=F32)
    # np.sum over a float32 CSR row: float32 accumulation of small integers (exact)
    np.add.at(degree, row, val)
    degree = np.maximum(F32(1.), degree).astype(F32)
    d_inv = np.power(degree, F32(-0.5)).astype(F32)
    nval = ((d_inv[row] * val).astype(F32) * d_inv[col]).astype(F32)
    return row, col, nval


def spmm_coo(row, col, val, x, n_rows=None):
    """Y = M @ X for COO M — restated gspmm('mul','sum').  model.py:102/430/442."""
    x = np.asarray(x, dtype=F32)
    n_rows = int(n_rows if n_rows is not None else x.shape[0])
    y = np.zeros((n_rows, x.shape[1]), dtype=F32)
    np.add.at(y, row, x[col] * np.asarray(val, dtype=F32)[:, None])
    return y


def spmm_coo_f64(row, col, val, x, n_rows=None):
    """float64 version of ``spmm_coo`` (order-independent yardstick)."""
    x = np.asarray(x, dtype=np.float64)
    n_rows = int(n_rows if n_rows is not None else x.shape[0])
    y = np.zeros((n_rows, x.shape[1]), dtype=np.float64)
    np.add.at(y, row, x[col] * np.asarray(val, dtype=np.float64)[:, None])
    return y


# --------------------------------------------------------------------------
# LightGCN
# --------------------------------------------------------------------------
def lightgcn_get_rep(adj, emb, n_layers):
    """mean(X_0 .. X_K), X_{l+1} = A_hat X_l.  model.py:96-106."""
    row, col, val = adj
    x = np.asarray(emb, dtype=F32)
    layers = [x]
    for _ in range(n_layers):
        x = spmm_coo(row, col, val, x)
        layers.append(x)
    return np.stack(layers, axis=0).mean(axis=0, dtype=F32)


def bpr_forward_lightgcn(rep, emb, n_users, users, pos, neg):
    """model.py:108-116 — l2 is of the RAW embedding rows."""
    users, pos, neg = (np.asarray(a, dtype=np.int64) for a in (users, pos, neg))
    ue, pe, ne = emb[users], emb[n_users + pos], emb[n_users + neg]
    l2 = (ue ** 2).sum(1) + (pe ** 2).sum(1) + (ne ** 2).sum(1)
    return rep[users], rep[n_users + pos], rep[n_users + neg], l2.astype(F32)


def bpr_forward_rep(rep, n_users, users, pos, neg):
    """NGCF.bpr_forward as used by IGCN/IMF (model.py:293-299, :448-449) —
    l2 is of the PROPAGATED rows."""
    users, pos, neg = (np.asarray(a, dtype=np.int64) for a in (users, pos, neg))
    ur, pr, nr = rep[users], rep[n_users + pos], rep[n_users + neg]
    l2 = (ur ** 2).sum(1) + (pr ** 2).sum(1) + (nr ** 2).sum(1)
    return ur, pr, nr, l2.astype(F32)


def bpr_forward_mf(user_emb, item_emb, users, pos, neg):
    """MF.bpr_forward.  model.py:62-67."""
    users, pos, neg = (np.asarray(a, dtype=np.int64) for a in (users, pos, neg))
    ue, pe, ne = user_emb[users], item_emb[pos], item_emb[neg]
    l2 = (ue ** 2).sum(1) + (pe ** 2).sum(1) + (ne ** 2).sum(1)
    return ue, pe, ne, l2.astype(F32)


def softplus(x):
    """torch.nn.functional.softplus, beta=1, threshold=20."""
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > 20., x, np.log1p(np.exp(np.minimum(x, 20.))))


def bpr_loss(users_r, pos_r, neg_r, l2_norm_sq, l2_reg, w=None):
    """trainer.py:238-243 (and :309-311 with ``w``).  Returns (bpr, reg)."""
    if w is None:
        pos_s = (users_r * pos_r).sum(1)
        neg_s = (users_r * neg_r).sum(1)
    else:
        pos_s = (users_r * pos_r * w[None, :]).sum(1)
        neg_s = (users_r * neg_r * w[None, :]).sum(1)
    bpr = softplus(neg_s.astype(np.float64) - pos_s.astype(np.float64)).mean()
    reg = 0. if l2_norm_sq is None else l2_reg * float(np.mean(l2_norm_sq, dtype=np.float64))
    return float(bpr), float(reg)


def bpr_grads(users_r, pos_r, neg_r):
    """d(mean softplus(neg-pos)) / d(users_r, pos_r, neg_r) in float64."""
    u, p, n = (np.asarray(a, dtype=np.float64) for a in (users_r, pos_r, neg_r))
    x = (u * n).sum(1) - (u * p).sum(1)
    sig = 1. / (1. + np.exp(-x))
    c = (sig / u.shape[0])[:, None]
    return c * (n - p), -c * u, c * u


def predict(rep, n_users, users):
    """rep[users] @ rep[n_users:].T.  model.py:118-123."""
    users = np.asarray(users, dtype=np.int64)
    return rep[users].astype(F32) @ rep[n_users:].astype(F32).T


# --------------------------------------------------------------------------
# INMO (IGCN / IMF)
# --------------------------------------------------------------------------
def graph_rank_nodes(train_array, n_users, n_items, ranking_metric):
    """Template ranking, 'degree' and 'sort'.  utils.py:94-123."""
    row, col, val = generate_adj(train_array, n_users, n_items)
    n = n_users + n_items
    if ranking_metric == 'degree':
        deg = np.zeros(n, dtype=F32)
        np.add.at(deg, row, val)
        user_metrics, item_metrics = deg[:n_users], deg[n_users:]
    elif ranking_metric in ('sort', 'greedy'):
        # sklearn normalize(axis=1, norm='l1') then column sums (utils.py:111-113)
        rs = np.zeros(n, dtype=F32)
        np.add.at(rs, row, np.abs(val))
        rs[rs == 0] = 1.
        nv = (val / rs[row]).astype(F32)
        cs = np.zeros(n, dtype=F32)
        np.add.at(cs, col, nv)
        user_metrics, item_metrics = cs[:n_users], cs[n_users:]
    else:
        return None
    return np.argsort(user_metrics)[::-1].copy(), np.argsort(item_metrics)[::-1].copy()


def igcn_generate_feat(train_array, n_users, n_items, user_map=None, item_map=None):
    """Template feature matrix F.  model.py:386-421.

    ``user_map`` / ``item_map``: dict original id -> template id (None = all,
    i.e. feature_ratio = 1, model.py:392-394).  Returns
    (row, col, ones, row_sum, user_map, item_map, shape).
    """
    if user_map is None:
        user_map = {u: u for u in range(n_users)}
    if item_map is None:
        item_map = {i: i for i in range(n_items)}
    user_dim, item_dim = len(user_map), len(item_map)
    rows, cols = [], []
    for user, item in np.asarray(train_array, dtype=np.int64).reshape(-1, 2).tolist():
        if item in item_map:                                   # model.py:409-410
            rows.append(user); cols.append(user_dim + item_map[item])
        if user in user_map:                                   # model.py:411-412
            rows.append(n_users + item); cols.append(user_map[user])
    for user in range(n_users):                                # model.py:413-414
        rows.append(user); cols.append(user_dim + item_dim)
    for item in range(n_items):                                # model.py:415-416
        rows.append(n_users + item); cols.append(user_dim + item_dim + 1)
    shape = (n_users + n_items, user_dim + item_dim + 2)
    r, c, v = coo_sum_duplicates(rows, cols, np.ones(len(rows), dtype=F32), shape)
    row_sum = np.zeros(shape[0], dtype=F32)
    np.add.at(row_sum, r, v)
    return r, c, v, row_sum, user_map, item_map, shape


def igcn_feat_values(feat_row, row_sum, alpha):
    """val[e] = row_sum[row[e]] ** ((alpha-1)/2 - 0.5).  model.py:374-377."""
    expo = (alpha - 1.) / 2. - 0.5
    return np.power(row_sum[feat_row].astype(F32), F32(expo)).astype(F32)


def dropout_keep_scale(values, keep_mask, p):
    """model.py:263-275 given an explicit keep mask: kept values / (1 - p)."""
    out = np.zeros_like(values, dtype=F32)
    out[keep_mask] = (values[keep_mask] / F32(1. - p)).astype(F32)
    return out


def igcn_get_rep(adj, feat, feat_val, template_emb, n_layers, imf=False):
    """IGCN.get_rep (model.py:434-446) / IMF.get_rep (model.py:540-543), eval mode
    (or train mode when ``feat_val`` already carries the dropout mask/scale)."""
    frow, fcol, shape = feat
    x = spmm_coo(frow, fcol, feat_val, template_emb, n_rows=shape[0])   # model.py:423-432
    if imf:
        return x
    return lightgcn_get_rep(adj, x, n_layers)


# --------------------------------------------------------------------------
# evaluation
# --------------------------------------------------------------------------
def eval_topk(scores, exclude_lists=None, banned_items=None, k=20):
    """mask -> top-k.  trainer.py:149-164.  Ties broken by lower item id
    (torch.topk leaves tie order unspecified)."""
    s = np.array(scores, dtype=F32, copy=True)
    if exclude_lists is not None:
        for u, items in enumerate(exclude_lists):
            if len(items):
                s[u, np.asarray(items, dtype=np.int64)] = -np.inf
    if banned_items is not None:
        s[:, np.asarray(banned_items, dtype=np.int64)] = -np.inf
    order = np.lexsort((np.arange(s.shape[1])[None, :].repeat(s.shape[0], 0), -s), axis=1)
    return order[:, :k].astype(np.int64)


def calculate_metrics(eval_data, rec_items, topks):
    """Precision / Recall / NDCG @k.  trainer.py:109-138 (same dtypes)."""
    rec_items = np.asarray(rec_items)
    results = {'Precision': {}, 'Recall': {}, 'NDCG': {}}
    hit = np.zeros(rec_items.shape, dtype=F32)
    for u in range(rec_items.shape[0]):
        s = set(eval_data[u])
        for j in range(rec_items.shape[1]):
            if int(rec_items[u, j]) in s:
                hit[u, j] = 1.
    lens = np.array([len(x) for x in eval_data], dtype=np.int32)
    for k in topks:
        hit_num = np.sum(hit[:, :k], axis=1)
        precisions = hit_num / k
        with np.errstate(invalid='ignore', divide='ignore'):
            recalls = hit_num / lens
        max_hit_num = np.minimum(lens, k)
        max_hit = np.zeros_like(hit[:, :k], dtype=F32)
        for u, num in enumerate(max_hit_num):
            max_hit[u, :num] = 1.
        denom = np.log2(np.arange(2, k + 2, dtype=F32))[None, :]
        dcgs = np.sum(hit[:, :k] / denom, axis=1)
        idcgs = np.sum(max_hit / denom, axis=1)
        with np.errstate(invalid='ignore', divide='ignore'):
            ndcgs = dcgs / idcgs
        m = max_hit_num > 0
        results['Precision'][k] = precisions[m].mean()
        results['Recall'][k] = recalls[m].mean()
        results['NDCG'][k] = ndcgs[m].mean()
    return results


# --------------------------------------------------------------------------
# datasets
# --------------------------------------------------------------------------
def read_data(path):
    """'user item item ...' per line; returns (lists, n_items_seen).  dataset.py:154-164."""
    data, n_items = [], 0
    with open(path, 'r') as f:
        lines = f.read().strip().split('\n')
    for line in lines:
        items = [int(t) for t in line.split(' ')[1:]]
        if items:
            n_items = max(n_items, max(items) + 1)
        data.append(items)
    return data, n_items


def auxiliary_train_data(train_data, user_map, item_map):
    """AuxiliaryDataset re-indexing.  dataset.py:258-273."""
    out = [[] for _ in range(len(user_map))]
    for o_user in range(len(train_data)):
        if o_user in user_map:
            for o_item in train_data[o_user]:
                if o_item in item_map:
                    out[user_map[o_user]].append(item_map[o_item])
    return out

"""Host-side graph construction and the device CSR container.

Counterpart of the reference's utils.get_sparse_tensor / generate_daj_mat
(utils.py:32-49), LightGCN.generate_graph (model.py:85-94) and
IGCN.generate_feat (model.py:386-421).  The reference keeps a coalesced COO
tensor and rebuilds a DGL graph from it on every get_rep call
(model.py:99-100); here the matrix is converted ONCE into CSR resident in HBM,
together with the long-row schedule the SpMM kernel uses.

Everything here is vectorised numpy on the host; nothing is per-step.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

LONG_THRESHOLD = 256       # rows with more nonzeros are cut into segments (tuned on MI355X, profiles/)
SEGMENT_LEN = 256


def coo_to_csr_host(row, col, val, shape):
    """Row-major sorted CSR with duplicate entries summed (the result of the
    reference's coo_matrix(...).tocsr() + coalesce(), utils.py:32-38, :46-48).
    Returns (rowptr int64, col int32, val float32)."""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val, dtype=np.float32)
    n_rows, n_cols = int(shape[0]), int(shape[1])
    if row.size:
        key = row * np.int64(n_cols) + col
        order = np.argsort(key, kind='stable')
        key = key[order]
        first = np.empty(key.shape, dtype=bool)
        first[0] = True
        np.not_equal(key[1:], key[:-1], out=first[1:])
        starts = np.flatnonzero(first)
        val = np.add.reduceat(val[order], starts).astype(np.float32)
        key = key[starts]
        row, col = key // n_cols, key % n_cols
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=n_rows), out=rowptr[1:])
    return rowptr, col.astype(np.int32), val


def adjacency_host(train_array, n_users, n_items):
    """A = [[0, R], [R^T, 0]] as host CSR.  utils.py:41-49."""
    ta = np.asarray(train_array, dtype=np.int64).reshape(-1, 2)
    users, items = ta[:, 0], ta[:, 1] + n_users
    row = np.concatenate([users, items])
    col = np.concatenate([items, users])
    n = n_users + n_items
    return coo_to_csr_host(row, col, np.ones(row.shape, dtype=np.float32), (n, n))


def normalized_adjacency_host(train_array, n_users, n_items):
    """A_hat = D^-1/2 A D^-1/2, deg = max(1, rowsum), float32 in the reference's
    operation order (d_mat.dot(adj).dot(d_mat)).  model.py:85-94."""
    rowptr, col, val = adjacency_host(train_array, n_users, n_items)
    n = n_users + n_items
    row = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    degree = np.zeros(n, dtype=np.float32)
    np.add.at(degree, row, val)
    degree = np.maximum(np.float32(1.), degree).astype(np.float32)
    d_inv = np.power(degree, np.float32(-0.5)).astype(np.float32)
    val = ((d_inv[row] * val).astype(np.float32) * d_inv[col]).astype(np.float32)
    return rowptr, col, val


def transpose_host(rowptr, col, n_cols):
    """CSR of M^T plus, for every entry of M^T, the position of the same entry in
    M (edge id) so that both views drop the same edges under dropout."""
    n_rows = rowptr.shape[0] - 1
    row = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr))
    order = np.argsort(col.astype(np.int64) * np.int64(n_rows) + row, kind='stable')
    t_rowptr = np.zeros(n_cols + 1, dtype=np.int64)
    np.cumsum(np.bincount(col, minlength=n_cols), out=t_rowptr[1:])
    return t_rowptr, row[order].astype(np.int32), order.astype(np.int32)


def feature_matrix_host(train_array, n_users, n_items, user_map=None, item_map=None):
    """INMO template feature matrix F (binary structure) and row_sum.  model.py:386-421.

    user_map / item_map: dict original id -> template id, or None for "every
    node is a template" (feature_ratio = 1., model.py:392-394).
    Returns (rowptr, col, row_sum float32, shape)."""
    ta = np.asarray(train_array, dtype=np.int64).reshape(-1, 2)
    users, items = ta[:, 0], ta[:, 1]
    if user_map is None:
        u_lut = np.arange(n_users, dtype=np.int64)
        user_dim = n_users
    else:
        u_lut = np.full(n_users, -1, dtype=np.int64)
        if len(user_map):
            k = np.fromiter(user_map.keys(), dtype=np.int64, count=len(user_map))
            v = np.fromiter(user_map.values(), dtype=np.int64, count=len(user_map))
            keep = k < n_users
            u_lut[k[keep]] = v[keep]
        user_dim = len(user_map)
    if item_map is None:
        i_lut = np.arange(n_items, dtype=np.int64)
        item_dim = n_items
    else:
        i_lut = np.full(n_items, -1, dtype=np.int64)
        if len(item_map):
            k = np.fromiter(item_map.keys(), dtype=np.int64, count=len(item_map))
            v = np.fromiter(item_map.values(), dtype=np.int64, count=len(item_map))
            keep = k < n_items
            i_lut[k[keep]] = v[keep]
        item_dim = len(item_map)
    it, ut = i_lut[items], u_lut[users]
    m_i, m_u = it >= 0, ut >= 0
    row = np.concatenate([users[m_i], n_users + items[m_u],
                          np.arange(n_users, dtype=np.int64), n_users + np.arange(n_items, dtype=np.int64)])
    col = np.concatenate([user_dim + it[m_i], ut[m_u],
                          np.full(n_users, user_dim + item_dim, dtype=np.int64),
                          np.full(n_items, user_dim + item_dim + 1, dtype=np.int64)])
    shape = (n_users + n_items, user_dim + item_dim + 2)
    rowptr, col, val = coo_to_csr_host(row, col, np.ones(row.shape, dtype=np.float32), shape)
    row_sum = np.add.reduceat(val, rowptr[:-1]).astype(np.float32)   # every row has its global column
    return rowptr, col, row_sum, shape


def _csr_from_keys_device(key, n_rows, n_cols):
    """Row-major sorted CSR with duplicates counted, from int64 keys = row * n_cols + col on the device:
    (rowptr int64 [n_rows + 1], col int32, count float32) — the device form of coo_to_csr_host for unit values."""
    ukey, counts = torch.unique(key, sorted=True, return_counts=True)
    row = torch.div(ukey, n_cols, rounding_mode='floor')
    col = (ukey - row * n_cols).to(torch.int32)
    rowptr = torch.empty(n_rows + 1, dtype=torch.int64, device=key.device)
    _lib.check(_lib.lib().igcn_csr_from_sorted_coo(row.data_ptr(), row.numel(), n_rows, rowptr.data_ptr(), _lib.current_stream()),
               'igcn_csr_from_sorted_coo')
    return rowptr, row, col, counts.to(torch.float32)


def normalized_adjacency_device(train_array, n_users, n_items, device, xcd_plan='default'):
    """normalized_adjacency_host built in HBM (the graph-swap path of the inductive update, run/dropui/igcn_dropui.py:
    26-35, rebuilds the graph on a live model): one sort of the 2 T keys on the GPU instead of host argsorts.  Values
    are BIT-IDENTICAL to the host builder / model.py:85-94: degrees are small integers, so D^-1/2 comes from a table of
    numpy float32 powers, and the two float32 products are taken in the reference's order."""
    ta = torch.from_numpy(np.ascontiguousarray(np.asarray(train_array, dtype=np.int64).reshape(-1, 2))).to(device)
    n = n_users + n_items
    users, items = ta[:, 0], ta[:, 1] + n_users
    key = torch.cat([users * n + items, items * n + users])
    rowptr, row, col, val = _csr_from_keys_device(key, n, n)
    degree = torch.zeros(n, dtype=torch.float32, device=device).index_add_(0, row, val).clamp_(min=1.)
    max_deg = int(degree.max().item()) if n else 1
    table = np.power(np.maximum(np.arange(max_deg + 1, dtype=np.float32), np.float32(1.)), np.float32(-0.5)).astype(np.float32)
    d_inv = torch.from_numpy(table).to(device)[degree.to(torch.int64)]
    val = (d_inv[row] * val) * d_inv[col.to(torch.int64)]
    return CsrMatrix.from_device(rowptr, col, val, (n, n), order_blocks=[0, n_users, n],
                                 xcd_plan=XCD_PLAN if xcd_plan == 'default' else xcd_plan)


def feature_matrix_device(train_array, n_users, n_items, user_map, item_map, device):
    """feature_matrix_host built in HBM: (CsrMatrix F with implicit unit values, row_sum float32 tensor)."""
    ta = torch.from_numpy(np.ascontiguousarray(np.asarray(train_array, dtype=np.int64).reshape(-1, 2))).to(device)
    users, items = ta[:, 0], ta[:, 1]

    def lut(mapping, n):
        if mapping is None:
            return torch.arange(n, dtype=torch.int64, device=device), n
        t = np.full(n, -1, dtype=np.int64)
        if len(mapping):
            k = np.fromiter(mapping.keys(), dtype=np.int64, count=len(mapping))
            v = np.fromiter(mapping.values(), dtype=np.int64, count=len(mapping))
            keep = k < n
            t[k[keep]] = v[keep]
        return torch.from_numpy(t).to(device), len(mapping)
    u_lut, user_dim = lut(user_map, n_users)
    i_lut, item_dim = lut(item_map, n_items)
    it, ut = i_lut[items], u_lut[users]
    m_i, m_u = it >= 0, ut >= 0
    n_cols = user_dim + item_dim + 2
    ar_u = torch.arange(n_users, dtype=torch.int64, device=device)
    ar_i = torch.arange(n_items, dtype=torch.int64, device=device)
    key = torch.cat([users[m_i] * n_cols + (user_dim + it[m_i]), (n_users + items[m_u]) * n_cols + ut[m_u],
                     ar_u * n_cols + (user_dim + item_dim), (n_users + ar_i) * n_cols + (user_dim + item_dim + 1)])
    n_rows = n_users + n_items
    rowptr, row, col, val = _csr_from_keys_device(key, n_rows, n_cols)
    row_sum = torch.zeros(n_rows, dtype=torch.float32, device=device).index_add_(0, row, val)
    f = CsrMatrix.from_device(rowptr, col, None, (n_rows, n_cols), order_blocks=[0, n_users, n_rows], xcd_plan=XCD_PLAN_FEATURES)
    f.transposed_order_blocks = [0, user_dim, user_dim + item_dim, n_cols]
    return f, row_sum


def graph_rank_nodes(dataset, ranking_metric):
    """Template ranking for feature_ratio < 1 (utils.py:94-123): 'degree' = row sums
    of A, 'sort' / 'greedy' = column sums of the row-L1-normalised A.  Returns
    (ranked_users, ranked_items), best first, as the reference's argsort()[::-1].

    The metric is formed from the dataset's (user, item)-sorted train pairs without building A (round 4: the 2 T-entry
    sort of adjacency_host was 0.15 s of a 0.15 s call at Yelp size) but with the SAME float32 operations in the same
    order as scipy's column sum over the coalesced CSR — per target column the contributions arrive by ascending source
    row — so the values, and with them numpy's argsort order on ties, are bit-identical to the reference's
    (tests/test_host_cpu.py::test_rank_nodes_matches_reference, ::test_rank_metric_equals_the_adjacency_form)."""
    n_users, n_items = dataset.n_users, dataset.n_items
    if ranking_metric not in ('degree', 'sort', 'greedy'):
        raise ValueError("ranking_metric %r not supported (use 'degree' or 'sort')" % (ranking_metric,))
    if hasattr(dataset, 'csr'):
        rowptr, col = dataset.csr('train', sort=True)             # every user's items ascending (cached by the dataset)
        users = np.repeat(np.arange(n_users, dtype=np.int64), np.diff(rowptr))
        items = np.asarray(col, dtype=np.int64)
    else:                                                         # anything with the reference's attributes (utils.py:94 reads train_array)
        ta = np.asarray(dataset.train_array, dtype=np.int64).reshape(-1, 2)
        order = np.lexsort((ta[:, 1], ta[:, 0]))
        users, items = ta[order, 0], ta[order, 1]
    mult = np.ones(items.shape[0], dtype=np.float32)
    if items.size:                                                # duplicate pairs are summed by the reference's coo -> csr
        first = np.ones(items.shape[0], dtype=bool)
        first[1:] = (users[1:] != users[:-1]) | (items[1:] != items[:-1])
        if not first.all():
            starts = np.flatnonzero(first)
            mult = np.add.reduceat(mult, starts).astype(np.float32)
            users, items = users[starts], items[starts]
    deg_u = np.zeros(n_users, dtype=np.float32)
    deg_i = np.zeros(n_items, dtype=np.float32)
    np.add.at(deg_u, users, mult)                                 # small integers: exact in any order
    np.add.at(deg_i, items, mult)
    if ranking_metric == 'degree':
        metric_u, metric_i = deg_u, deg_i
    else:
        rs_u, rs_i = deg_u.copy(), deg_i.copy()
        rs_u[rs_u == 0] = 1.
        rs_i[rs_i == 0] = 1.
        metric_u = np.zeros(n_users, dtype=np.float32)
        metric_i = np.zeros(n_items, dtype=np.float32)
        # column n_users + i of A collects m / rs[u] from the user rows, u ascending; column u collects m / rs[i] from the
        # item rows, i ascending — the pair list is sorted by (u, i), so both sequences are the list's own order per target
        np.add.at(metric_i, items, (mult / rs_u[users]).astype(np.float32))
        np.add.at(metric_u, users, (mult / rs_i[items]).astype(np.float32))
    return np.argsort(metric_u)[::-1].copy(), np.argsort(metric_i)[::-1].copy()


def graph_rank_nodes_from_adjacency(dataset, ranking_metric):
    """The same ranking formed the long way round, from the coalesced adjacency matrix as the reference does (utils.py:94-123);
    kept as the cross-check of graph_rank_nodes."""
    n_users, n_items = dataset.n_users, dataset.n_items
    rowptr, col, val = adjacency_host(dataset.train_array, n_users, n_items)
    n = n_users + n_items
    row = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    if ranking_metric == 'degree':
        metric = np.zeros(n, dtype=np.float32)
        np.add.at(metric, row, val)
    elif ranking_metric in ('sort', 'greedy'):
        rs = np.zeros(n, dtype=np.float32)
        np.add.at(rs, row, np.abs(val))
        rs[rs == 0] = 1.
        metric = np.zeros(n, dtype=np.float32)
        np.add.at(metric, col, (val / rs[row]).astype(np.float32))
    else:
        raise ValueError("ranking_metric %r not supported (use 'degree' or 'sort')" % (ranking_metric,))
    return np.argsort(metric[:n_users])[::-1].copy(), np.argsort(metric[n_users:])[::-1].copy()


# Default plan of the propagation matrices (A_hat): rows above 112 nonzeros are cut at the slice boundaries.  Measured on
# MI355X, same-process A/Bs against the round-2 dealing order (profiles/r03b_*, r03d_*): Amazon-like d = 64 -11 %, d = 128
# -11 %, Yelp-like -16 %, Gowalla-like -26 %, d = 32 +-1 %; thresholds 96...160 within 2 % of each other, 64 and below lose
# (8 partial rows per cut row).  None = the plain long-row plan.
CLOSING_AT = 0.25              # where a phase's closing segments are dealt: this fraction into the phase's rows (xcd_plan / CsrMatrix._build_plan)
MIN_CLOSING_GAP = 8            # rows that must lie between a list's other segments and its closing ones for the plan to be foldable in the launch
XCD_PLAN = {'threshold': 112}
XCD_PLAN_FEATURES = XCD_PLAN   # INMO's template-feature matrix F and its transposed view: -1...-3 % (profiles/r03m_*); None = plain plan
N_XCD = 8                  # lists of the XCD plan = XCDs of an MI355X (igcn_spmm_csr_f32: xcd_off has N_XCD + 1 entries)


def xcd_plan(rowptr, col, blocks, slice_threshold, segment_len, row_cost=4, assign='affinity', list_order='segments_first', n_lists=None,
             closing_at=CLOSING_AT, late_every=0, cold_rows=0, info=None):
    """The XCD plan of a CSR matrix: the work of one SpMM launch cut into N_XCD lists, one per XCD, such that a list
    gathers as much as possible from ONE slice of the operand — a slice (1/8 of the operand's rows) fits an XCD's 4 MiB
    L2 where the whole operand does not, and the eight L2s are private (igcn_hip.h: xcd_off).

    Per block of rows (for A_hat: the user rows, which gather item rows, then the item rows, which gather user rows):
      * the block's column range is cut into N_XCD contiguous slices with equal numbers of gathers;
      * a row with more than `slice_threshold` nonzeros is cut at the slice boundaries (columns are ascending inside
        a row, so a slice is a contiguous run of its nonzeros) and each piece into segments of <= segment_len
        nonzeros; a segment goes to the list of its slice.  Segments write partial sums that the long-row reduce
        kernel adds up in slot order, exactly as for the plain long-row plan — up to 8 partial rows per cut row, which
        only pays for rows of a hundred nonzeros or so (measured: profiles/r03b_*, r03d_*);
      * a shorter row is computed whole by the list that holds most of its columns (assign='affinity'; 'spread' deals
        them without looking), as long as that list's share of the block's work (nonzeros + row_cost per row) is not
        exceeded; the overflow fills the lists that are short.
    Inside a list: per block the segments first, then the rows by descending length (igcn_hip.h: row_order).

    rowptr int64 [n_rows + 1], col int32/int64 [nnz]: torch tensors on ANY device — the plan is built where the matrix
    lives (in HBM for the device builders: the inductive update rebuilds the graph on a live model).
    Returns (long_rows int32 [n_long, 4], segments int32 [n_seg, 6] — the byte layouts of igcn_long_row /
    igcn_row_segment —, row_order int32, xcd_off int64 [N_XCD + 1], load float64 [N_XCD]) on that device.
    info (a dict, optional): receives 'foldable' — False when some list has fewer than MIN_CLOSING_GAP rows to put between a cut
    row's closing segment and its other segments (such a plan is never added up inside the launch)."""
    NL = int(n_lists or N_XCD)        # lists of the plan (developer A/Bs cut fewer slices; the kernel always walks N_XCD)
    if info is None:
        info = {}
    info['foldable'] = list_order == 'segments_first'     # (out-parameter: may this plan's cut rows be added up inside the launch?)
    dev = rowptr.device
    i64 = dict(dtype=torch.int64, device=dev)
    n_rows = rowptr.shape[0] - 1
    lens = rowptr[1:] - rowptr[:-1]
    if isinstance(slice_threshold, (list, tuple)):                 # one threshold per block of rows (developer A/Bs)
        thr_row = torch.empty(n_rows, **i64)
        for (lo, hi), t in zip(zip(blocks[:-1], blocks[1:]), slice_threshold):
            thr_row[lo:hi] = int(t)
        cut = lens > thr_row
    else:
        cut = lens > slice_threshold
    rp = rowptr.cpu().tolist() if len(blocks) <= 8 else None
    seg_parts, per_block = [], []
    empty = torch.zeros(0, **i64)
    for lo, hi in zip(blocks[:-1], blocks[1:]):
        e0, e1 = (rp[lo], rp[hi]) if rp is not None else (int(rowptr[lo]), int(rowptr[hi]))
        rows = torch.arange(lo, hi, **i64)
        whole = rows[~cut[lo:hi]]
        if e1 == e0:
            per_block.append((whole, torch.zeros_like(whole), torch.zeros(whole.shape[0], dtype=torch.float64, device=dev)))
            continue
        c = col[e0:e1].to(torch.int64)
        cmin = int(c.min())
        cum = torch.cumsum(torch.bincount(c - cmin), 0)
        targets = (torch.arange(1, NL, **i64).to(torch.float64) * (float(cum[-1]) / NL))
        bounds = cmin + 1 + torch.searchsorted(cum.to(torch.float64), targets)
        sl = torch.searchsorted(bounds, c, right=True)                 # slice of every nonzero of the block, 0..NL-1
        row_e = torch.repeat_interleave(rows, lens[lo:hi])
        in_cut = cut[row_e]
        pos = e0 + torch.nonzero(in_cut).flatten()                     # nonzeros of the cut rows, in storage order
        if pos.numel():
            key = row_e[in_cut] * NL + sl[in_cut]
            first = torch.ones_like(key, dtype=torch.bool)
            first[1:] = (key[1:] != key[:-1]) | (pos[1:] != pos[:-1] + 1)
            fidx = torch.nonzero(first).flatten()
            p_start, p_key = pos[fidx], key[fidx]
            p_len = torch.diff(torch.cat([fidx, torch.tensor([key.shape[0]], **i64)]))
            n_chunks = (p_len + segment_len - 1) // segment_len
            rep = torch.repeat_interleave(torch.arange(p_start.shape[0], **i64), n_chunks)
            within = torch.arange(rep.shape[0], **i64) - torch.repeat_interleave(torch.cumsum(n_chunks, 0) - n_chunks, n_chunks)
            s_start = p_start[rep] + within * segment_len
            s_len = torch.clamp(p_len[rep] - within * segment_len, max=segment_len)
            seg_parts.append((s_start, s_len, p_key[rep] // NL, p_key[rep] % NL))
        # rows that stay whole: nonzeros per slice -> the list they would like and how much of the row it holds
        w_e = ~in_cut
        cnt = torch.bincount((row_e[w_e] - lo) * NL + sl[w_e], minlength=(hi - lo) * NL).reshape(hi - lo, NL)
        cnt = cnt[~cut[lo:hi]]
        best, pref = cnt.max(dim=1)
        if cold_rows:
            # developer A/B (round 6): prefer the list that holds most of the row's COLD columns — the `cold_rows` most-gathered
            # operand rows of the block are in every XCD's L2 whatever list gathers them, so only the others can gain from being
            # gathered by their slice's XCD; a row without cold columns keeps the plain preference
            per_col = torch.bincount(c - cmin)
            if per_col.shape[0] > cold_rows:
                hot_min = torch.sort(per_col, descending=True).values[cold_rows - 1]
                cold_e = w_e & (per_col[c - cmin] < hot_min)
                ccnt = torch.bincount((row_e[cold_e] - lo) * NL + sl[cold_e], minlength=(hi - lo) * NL).reshape(hi - lo, NL)
                ccnt = ccnt[~cut[lo:hi]]
                cbest, cpref = ccnt.max(dim=1)
                pref = torch.where(cbest > 0, cpref, pref)
                best = torch.where(cbest > 0, cnt.gather(1, cpref[:, None])[:, 0], best)
        per_block.append((whole, pref, best.to(torch.float64) / torch.clamp(lens[whole], min=1).to(torch.float64)))
    if seg_parts:
        seg_start, seg_len, seg_row, seg_xcd = (torch.cat([p[j] for p in seg_parts]) for j in range(4))
    else:
        seg_start = seg_len = seg_row = seg_xcd = empty
    n_seg = seg_start.shape[0]
    segments = torch.zeros((n_seg, 6), dtype=torch.int32, device=dev)
    if n_seg:
        segments[:, 0:2] = seg_start.contiguous().view(torch.int32).reshape(n_seg, 2)      # int64 start, little endian
        segments[:, 2], segments[:, 3], segments[:, 4] = seg_len.int(), torch.arange(n_seg, device=dev).int(), seg_row.int()
    long_ids = torch.nonzero(cut).flatten()
    is_closing = torch.zeros(n_seg, dtype=torch.bool, device=dev)
    long_rows = torch.zeros((long_ids.shape[0], 4), dtype=torch.int32, device=dev)
    if long_ids.shape[0]:
        # segments were generated block by block in storage order: those of one row are consecutive
        is_first = torch.ones(n_seg, dtype=torch.bool, device=dev)
        is_first[1:] = seg_row[1:] != seg_row[:-1]
        firsts = torch.nonzero(is_first).flatten()
        if firsts.shape[0] != long_ids.shape[0] or not torch.equal(seg_row[firsts], long_ids):
            raise ValueError('every cut row needs its own run of segments (rows sorted, nonzeros stored row by row)')
        long_rows[:, 0], long_rows[:, 1] = long_ids.int(), firsts.int()
        long_rows[:, 2] = torch.diff(torch.cat([firsts, torch.tensor([n_seg], **i64)])).int()
        # igcn_row_segment.long_index: the segment's row as an entry of long_rows (the launch folds a cut row through it);
        # (the arrival counters of the fold live behind the partial sums: CsrMatrix.partial)
        long_of_seg = torch.cumsum(is_first.to(torch.int64), 0) - 1
        is_closing = torch.ones(n_seg, dtype=torch.bool, device=dev)               # a row's LAST segment closes it (bit 31)
        is_closing[:-1] = seg_row[1:] != seg_row[:-1]
        packed = long_of_seg | (is_closing.to(torch.int64) << 31)
        segments[:, 5] = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)   # (bit 31 of an int32)

    # segments dealt late in their lists: the closing ones (developer sweeps: also every late_every-th segment of a row, counted from its end)
    is_late = is_closing
    if late_every and long_ids.shape[0]:
        from_end = (long_rows[:, 1] + long_rows[:, 2] - 1).to(torch.int64)[long_of_seg] - torch.arange(n_seg, **i64)
        is_late = is_closing | (from_end % int(late_every) == 0)
    # deal the blocks, one after the other, keeping the lists' total work level
    load = torch.zeros(NL, dtype=torch.float64, device=dev)
    lists = [[] for _ in range(NL)]
    inner = torch.tensor(list(blocks[1:-1]), **i64)
    seg_block = torch.searchsorted(inner, seg_row, right=True) if n_seg else seg_row
    for b, (whole, pref, aff) in enumerate(per_block):
        sb = torch.nonzero(seg_block == b).flatten() if n_seg else empty
        seg_cost = torch.bincount(seg_xcd[sb], weights=(seg_len[sb] + row_cost).to(torch.float64), minlength=NL) \
            if sb.numel() else torch.zeros(NL, dtype=torch.float64, device=dev)
        cost = (lens[whole] + row_cost).to(torch.float64)
        level = (load.sum() + seg_cost.sum() + cost.sum()) / NL
        quota = torch.clamp(level - load - seg_cost, min=0.)
        if float(quota.sum()) > 0:
            quota = quota * (cost.sum() / quota.sum())
        owner = torch.full((whole.shape[0],), -1, **i64)
        if assign == 'affinity' and whole.numel():
            order = torch.sort(-aff, stable=True).indices
            order = order[torch.sort(pref[order], stable=True).indices]      # by list, strongest affinity first
            po = pref[order]
            run = torch.cumsum(cost[order], 0)
            before = torch.bincount(po, weights=cost[order], minlength=NL).cumsum(0) - \
                torch.bincount(po, weights=cost[order], minlength=NL)           # work of the lists before po
            fits = (run - before[po]) <= quota[po]
            owner[order[fits]] = po[fits]
        rest = torch.nonzero(owner < 0).flatten()
        if rest.numel():
            rest = rest[torch.sort(-lens[whole[rest]], stable=True).indices]
            got = owner >= 0
            used = torch.bincount(owner[got], weights=cost[got], minlength=NL) if bool(got.any()) else torch.zeros_like(quota)
            room = torch.clamp(quota - used, min=0.)
            edges = torch.cumsum(room * (cost[rest].sum() / torch.clamp(room.sum(), min=1e-30)), 0)
            mid = torch.cumsum(cost[rest], 0) - cost[rest] / 2
            owner[rest] = torch.clamp(torch.searchsorted(edges, mid), max=NL - 1)
        for x in range(NL):
            sx = sb[seg_xcd[sb] == x]
            sx = n_rows + sx[torch.sort(-seg_len[sx], stable=True).indices]
            rx = whole[owner == x]
            rx = rx[torch.sort(-lens[rx], stable=True).indices]
            if list_order == 'segments_first' and sx.numel():
                # closing segments (the launch adds a cut row up when its closing segment is reached: igcn_hip.h) a quarter into the
                # rows of the phase (CLOSING_AT) — the phase's rows come by descending length, so by then a good third of its time has
                # passed and the row's other segments, first in every list, are done; later (half-way, the first choice) these
                # full-length segments with a fold behind them end AFTER the list's last short rows: +12 us of tail (profiles/r05c_*)
                cl = is_late[sx - n_rows]
                at = int(rx.shape[0] * closing_at)
                if bool(cl.any()) and at < MIN_CLOSING_GAP:
                    # too few rows to put between a row's closing segment and the other segments: a wave could meet a closing segment
                    # and one of its siblings in the SAME visit (up to 4 entries), or before the siblings' waves have started — such a
                    # plan is never folded (the launch keeps the second kernel, whatever "spmm_fold" says)
                    info['foldable'] = False
                lists[x] += [sx[~cl], rx[:at], sx[cl], rx[at:]]
            elif list_order == 'rows_first':
                lists[x] += [rx, sx]
            elif list_order == 'interleaved' and sx.numel() and rx.numel():
                # spread the segments evenly among the rows (both keep their own order)
                pos_s = (torch.arange(sx.numel(), **i64).to(torch.float64) + 0.5) * ((sx.numel() + rx.numel()) / sx.numel())
                pos_r = (torch.arange(rx.numel(), **i64).to(torch.float64) + 0.5) * ((sx.numel() + rx.numel()) / rx.numel())
                both = torch.cat([sx, rx])[torch.sort(torch.cat([pos_s, pos_r]), stable=True).indices]
                lists[x].append(both)
            else:
                lists[x] += [sx, rx]
        load = load + seg_cost + (torch.bincount(owner, weights=cost, minlength=NL) if whole.numel() else 0.)
    per_list = [torch.cat(l) if l else empty for l in lists]
    # the kernels always walk N_XCD lists (xcd_off[(blockIdx.x & 7) + 1]): fewer lists of a developer A/B are padded with empty ones
    xcd_off = torch.zeros(max(NL, N_XCD) + 1, **i64)
    xcd_off[1:NL + 1] = torch.cumsum(torch.tensor([a.shape[0] for a in per_list], **i64), 0)
    xcd_off[NL + 1:] = xcd_off[NL]
    row_order = torch.cat(per_list).to(torch.int32)
    if row_order.shape[0] != n_rows - long_ids.shape[0] + n_seg:
        raise AssertionError('XCD plan lost or duplicated work items')
    return long_rows, segments, row_order, xcd_off, load


class CsrMatrix:
    """A CSR matrix resident in HBM with the SpMM long-row schedule.

    rowptr int64 [n_rows+1], col int32 [nnz], val float32 [nnz] or None (all
    ones), edge_id int32 [nnz] or None."""

    def __init__(self, rowptr, col, val, shape, device, edge_id=None,
                 long_threshold=LONG_THRESHOLD, segment_len=SEGMENT_LEN, keep_host=False, order_blocks=None, xcd_plan=None):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        self.shape = (int(shape[0]), int(shape[1]))
        if rowptr.shape[0] != self.shape[0] + 1 or rowptr[0] != 0 or rowptr[-1] != col.shape[0]:
            raise ValueError('inconsistent CSR arrays')
        if col.size and (col.min() < 0 or col.max() >= self.shape[1]):
            raise ValueError('column index out of range')
        self.nnz = int(col.shape[0])
        self.device = torch.device(device)
        self.rowptr_host = rowptr
        self.rowptr = torch.from_numpy(rowptr).to(self.device)
        self.col = torch.from_numpy(col).to(self.device)
        self.val = None if val is None else torch.from_numpy(np.ascontiguousarray(val, dtype=np.float32)).to(self.device)
        self.edge_id = None if edge_id is None else \
            torch.from_numpy(np.ascontiguousarray(edge_id, dtype=np.int32)).to(self.device)
        self.long_threshold = int(long_threshold)
        self.segment_len = int(segment_len)
        self.order_blocks = order_blocks
        self.xcd_plan = xcd_plan
        self._build_plan()
        self._partial = {}
        self._col_host = col if keep_host else None
        self._transposed = None

    @classmethod
    def from_device(cls, rowptr, col, val, shape, edge_id=None, long_threshold=LONG_THRESHOLD, segment_len=SEGMENT_LEN,
                    order_blocks=None, xcd_plan=None):
        """Wrap CSR arrays that already live in HBM (int64 rowptr, int32 col, float32 val or None);
        only rowptr is copied to the host, for the long-row schedule."""
        self = cls.__new__(cls)
        self.shape = (int(shape[0]), int(shape[1]))
        self.device = rowptr.device
        self.rowptr, self.col, self.val, self.edge_id = rowptr, col, val, edge_id
        self.nnz = int(col.shape[0])
        self.rowptr_host = rowptr.cpu().numpy()
        if self.rowptr_host.shape[0] != self.shape[0] + 1 or self.rowptr_host[-1] != self.nnz:
            raise ValueError('inconsistent CSR arrays')
        self.long_threshold, self.segment_len = int(long_threshold), int(segment_len)
        self.order_blocks = order_blocks
        self.xcd_plan = xcd_plan
        self._build_plan()
        self._partial = {}
        self._col_host = None
        self._transposed = None
        return self

    def transposed_view(self):
        """CSR of M^T (values all ones) sharing edge ids with M; built on first use, on the
        device (igcn_csr_transpose) when the matrix lives there."""
        if self._transposed is None:
            if self.device.type == 'cuda':
                L = _lib.lib()
                ws_bytes = L.igcn_csr_transpose_workspace_bytes(self.nnz)
                if ws_bytes < 0:
                    raise _lib.IgcnError('matrix too large for igcn_csr_transpose (nnz must be < 2^31)')
                ws = torch.empty(max(int(ws_bytes), 256), dtype=torch.uint8, device=self.device)
                t_rowptr = torch.empty(self.shape[1] + 1, dtype=torch.int64, device=self.device)
                t_col = torch.empty(self.nnz, dtype=torch.int32, device=self.device)
                edge_id = torch.empty(self.nnz, dtype=torch.int32, device=self.device)
                _lib.check(L.igcn_csr_transpose(self.rowptr.data_ptr(), _lib.ptr(self.col), self.shape[0], self.shape[1], self.nnz,
                                                t_rowptr.data_ptr(), _lib.ptr(t_col), _lib.ptr(edge_id), ws.data_ptr(),
                                                _lib.current_stream()), 'igcn_csr_transpose')
                self._transposed = CsrMatrix.from_device(t_rowptr, t_col, None, (self.shape[1], self.shape[0]), edge_id=edge_id,
                                                         long_threshold=LONG_THRESHOLD if self.xcd_plan else self.long_threshold,
                                                         segment_len=SEGMENT_LEN if self.xcd_plan else self.segment_len,
                                                         order_blocks=getattr(self, 'transposed_order_blocks', None),
                                                         xcd_plan=self.xcd_plan if getattr(self, 'transposed_order_blocks', None) else None)
            else:
                col = self._col_host if self._col_host is not None else self.col.cpu().numpy()
                self._transposed = CsrMatrix.transposed(self.rowptr_host, col, self.shape, self.device,
                                                        long_threshold=self.long_threshold, segment_len=self.segment_len)
            self._col_host = None
        return self._transposed

    def _build_plan(self):
        """xcd_plan: None, or {'threshold': T[, 'row_cost': c, 'assign': 'affinity' | 'spread']} — the XCD plan
        (xcd_plan_host) instead of the plain long-row plan + dealing order; needs order_blocks."""
        self.xcd_off = None
        if self.xcd_plan is not None and self.order_blocks is not None and self.nnz > 0:
            cfg = dict(self.xcd_plan)
            thr = cfg['threshold']
            self.long_threshold = int(min(thr)) if isinstance(thr, (list, tuple)) else int(thr)
            self.segment_len = min(self.segment_len, int(cfg.get('segment_len', self.long_threshold)))
            info = {}
            lr, sg, order, xcd_off, load = xcd_plan(self.rowptr, self.col, list(self.order_blocks), thr,
                                                    self.segment_len, cfg.get('row_cost', 4), cfg.get('assign', 'affinity'),
                                                    cfg.get('list_order', 'segments_first'), closing_at=cfg.get('closing_at', CLOSING_AT),
                                                    late_every=cfg.get('late_every', 0), cold_rows=cfg.get('cold_rows', 0), info=info)
            self.n_long, self.n_segments = int(lr.shape[0]), int(sg.shape[0])
            self.long_rows = lr.contiguous().view(torch.uint8).reshape(-1) if self.n_long else None
            self.segments = sg.contiguous().view(torch.uint8).reshape(-1) if self.n_long else None
            self.row_order, self.xcd_off, self.xcd_load = order, xcd_off, load
            self.closing_segments = bool(info['foldable']) and self.n_long > 0
            return
        L = _lib.lib()
        n_long, n_seg = C.c_int64(0), C.c_int64(0)
        _lib.check(L.igcn_spmm_plan_count_host(self.rowptr_host.ctypes.data, self.shape[0], self.long_threshold,
                                               self.segment_len, C.byref(n_long), C.byref(n_seg)), 'spmm_plan_count')
        self.n_long, self.n_segments = int(n_long.value), int(n_seg.value)
        self.long_rows = self.segments = None
        sg = None
        if self.n_long:
            lr = np.zeros(self.n_long, dtype=_lib.LONG_ROW_DTYPE)
            sg = np.zeros(self.n_segments, dtype=_lib.ROW_SEGMENT_DTYPE)
            _lib.check(L.igcn_spmm_plan_fill_host(self.rowptr_host.ctypes.data, self.shape[0], self.long_threshold,
                                                  self.segment_len, lr.ctypes.data, self.n_long,
                                                  sg.ctypes.data, self.n_segments), 'spmm_plan_fill')
            self.long_rows = torch.from_numpy(lr.view(np.uint8)).to(self.device)
            self.segments = torch.from_numpy(sg.view(np.uint8)).to(self.device)
        # The order in which rows and long-row segments are dealt to the waves.  Inside each block of rows (for A_hat:
        # the user rows, then the item rows, so that the launch still gathers from one table at a time): first the
        # segments of the block's long rows — the heaviest work items start first instead of forming the kernel's
        # tail — then the block's rows by descending length, so that the rows a wave works on together and the
        # waves next to it carry equal work.
        self.row_order = None
        self.closing_segments = False
        if self.order_blocks is not None and self.shape[0] > 0:
            n_rows = self.shape[0]
            lens = np.diff(self.rowptr_host)
            seg_row = sg['row'].astype(np.int64) if self.n_long else np.zeros(0, dtype=np.int64)
            closing = sg['long_index'] < 0 if self.n_long else np.zeros(0, dtype=bool)      # (bit 31: the row's closing segment)
            pieces = []
            bounds = list(self.order_blocks)
            foldable = True
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                in_block = (seg_row >= lo) & (seg_row < hi)
                rows = lo + np.argsort(-lens[lo:hi], kind='stable')
                # the block's segments first, its closing segments (they add their rows up, igcn_hip.h) CLOSING_AT into its rows
                at = int(len(rows) * CLOSING_AT)
                if at < MIN_CLOSING_GAP and bool((in_block & closing).any()):
                    foldable = False              # (see xcd_plan: a closing segment next to its siblings — the launch keeps the second kernel)
                pieces += [n_rows + np.flatnonzero(in_block & ~closing), rows[:at], n_rows + np.flatnonzero(in_block & closing), rows[at:]]
            self.closing_segments = self.n_long > 0 and foldable
            order = np.concatenate(pieces).astype(np.int32)
            assert order.shape[0] == n_rows + self.n_segments
            self.row_order = torch.from_numpy(order).to(self.device)

    def partial(self, d):
        """Workspace for the partial sums of long-row segments (n_segments x d)."""
        if not self.n_segments:
            return None
        buf = self._partial.get(d)
        if buf is None:
            # [n_segments, d] partial sums, then one arrival counter per cut row, 128 bytes apart (igcn_hip.h: zero when handed to the
            # library, zero again after every launch)
            buf = torch.empty(self.n_segments * d + self.n_long * 32, dtype=torch.float32, device=self.device)
            buf[self.n_segments * d:].zero_()
            self._partial[d] = buf
        return buf

    @classmethod
    def transposed(cls, rowptr, col, shape, device, **kw):
        t_rowptr, t_col, edge_id = transpose_host(np.asarray(rowptr), np.asarray(col), int(shape[1]))
        return cls(t_rowptr, t_col, None, (shape[1], shape[0]), device, edge_id=edge_id, **kw)

    def to_torch_coo(self):
        """Coalesced torch COO view (what the reference's norm_adj / feat_mat are)."""
        row = torch.repeat_interleave(torch.arange(self.shape[0], device=self.device), self.rowptr[1:] - self.rowptr[:-1])
        val = self.val if self.val is not None else torch.ones(self.nnz, device=self.device)
        return torch.sparse_coo_tensor(torch.stack([row, self.col.long()]), val, self.shape).coalesce()

"""Randomised sweeps of the two main kernels against the oracle (seeded; ~60 cases each)."""

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_spmm_fuzz_shapes_degrees_and_plans():
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    rng = np.random.default_rng(2024)
    for case in range(60):
        n_rows = int(rng.integers(1, 700))
        n_cols = int(rng.integers(1, 900))
        d = int(rng.choice([4, 8, 12, 16, 20, 32, 48, 64, 96, 128, 160, 256]))
        style = case % 4
        if style == 0:
            degs = rng.integers(0, min(n_cols, 40) + 1, size=n_rows)
        elif style == 1:                                   # heavy tail
            degs = np.minimum((rng.pareto(1.0, size=n_rows) * 3).astype(np.int64), n_cols)
        elif style == 2:                                   # lengths around the chunk / group boundaries
            degs = rng.choice([0, 1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257], size=n_rows)
            degs = np.minimum(degs, n_cols)
        else:                                              # a few very long rows
            degs = rng.integers(0, 6, size=n_rows)
            degs[rng.integers(0, n_rows, size=3)] = n_cols
        rowptr = np.zeros(n_rows + 1, dtype=np.int64)
        np.cumsum(degs, out=rowptr[1:])
        col = np.concatenate([np.sort(rng.choice(n_cols, size=int(k), replace=False)) for k in degs] + [np.zeros(0, np.int64)]).astype(np.int32)
        val = rng.standard_normal(col.shape[0]).astype(np.float32) if case % 3 else None
        lt = int(rng.choice([16, 64, 256, 1024]))
        sl = int(rng.choice([s for s in (16, 64, 256) if s <= lt]))
        csr = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda', long_threshold=lt, segment_len=sl)
        x = rng.standard_normal((n_cols, d)).astype(np.float32)
        n_adds = int(rng.integers(0, 4))
        adds = [rng.standard_normal((n_rows, d)).astype(np.float32) for _ in range(n_adds)]
        out_scale, add_scale = float(rng.uniform(0.2, 2)), float(rng.uniform(0.2, 2))
        y = spmm(csr, torch.from_numpy(x).cuda(), adds=[torch.from_numpy(a).cuda() for a in adds], out_scale=out_scale,
                 add_scale=add_scale).cpu().numpy()
        row = np.repeat(np.arange(n_rows, dtype=np.int64), degs)
        ref = out_scale * O.spmm_coo_f64(row, col.astype(np.int64), val if val is not None else np.ones(col.shape[0], np.float32), x, n_rows)
        for a in adds:
            ref = ref + add_scale * a.astype(np.float64)
        err = np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30)
        assert err < 1e-4, (case, n_rows, n_cols, d, style, lt, sl, err)


def test_score_topk_fuzz():
    from igcn_cf_amd import _lib
    from igcn_cf_amd.ops import score_topk
    rng = np.random.default_rng(77)
    for case in range(60):
        n_users = int(rng.integers(1, 400))
        n_items = int(rng.integers(1, 3000))
        d = int(rng.choice([4, 8, 16, 20, 32, 64, 64, 64, 100, 128]))
        k = int(rng.integers(1, min(n_items, 64) + 1))
        if case % 3 == 1:
            # few wave slots: every wave sweeps whole groups AND a run of the leftover groups' tiles (+ merge)
            _lib.set_tuning('topk_slots', int(rng.integers(1, 8)))
        else:
            _lib.set_tuning('topk_slots', None)
        U = rng.integers(-4, 5, size=(n_users, d)).astype(np.float32)             # exact dot products
        I = rng.integers(-4, 5, size=(n_items, d)).astype(np.float32)
        scores = U @ I.T
        kw, ex, ban = {}, None, None
        if case % 2 and n_items > k + 2:
            ex = [sorted(rng.choice(n_items, size=int(rng.integers(0, min(25, n_items - k))), replace=False).tolist())
                  for _ in range(n_users)]
            rowptr = np.zeros(n_users + 1, dtype=np.int64)
            np.cumsum([len(e) for e in ex], out=rowptr[1:])
            colx = np.array([i for e in ex for i in e], dtype=np.int32)
            kw.update(excl_rowptr=torch.from_numpy(rowptr).cuda(), excl_col=torch.from_numpy(colx).cuda())
        if case % 3 == 0 and n_items > k + 5:
            ban = np.sort(rng.choice(n_items, size=max(1, (n_items - k) // 3), replace=False))
            bm = np.zeros(n_items, dtype=np.uint8); bm[ban] = 1
            kw['banned'] = torch.from_numpy(bm).cuda()
        ids = rng.permutation(n_users).astype(np.int64)
        # d = 64 / 128, k <= 60: every other such case through the two-stage path (fp16 candidate sweep + exact re-scoring)
        mode = 'fast' if d in (64, 128) and k <= 60 and case % 2 == 0 else 'exact'
        # two bf16 planes each side / one fp16 item plane + two user planes / one fp16 plane each side (default)
        _lib.set_tuning('topk_fast_mode', 1 if case % 4 == 0 else 2 if case % 8 == 2 else None)
        _lib.set_tuning('topk_fast_narrow', 0 if case % 16 == 6 else None)           # 64-user wave-groups on a small batch
        _lib.set_tuning('topk_fast_share', 0 if case % 16 == 14 else None)          # pieces keep their thresholds to themselves
        idx, val = score_topk(torch.from_numpy(U).cuda(), torch.from_numpy(I).cuda(), k, user_ids=torch.from_numpy(ids).cuda(),
                              mode=mode, **kw)
        s = scores[ids].copy()
        if ex is not None:
            for b, u in enumerate(ids):
                if len(ex[u]):
                    s[b, np.asarray(ex[u])] = -np.inf
        if ban is not None:
            s[:, ban] = -np.inf
        ref = O.eval_topk(s, None, None, k=k)
        np.testing.assert_array_equal(idx.cpu().numpy(), ref, err_msg=str((case, n_users, n_items, d, k)))
        np.testing.assert_array_equal(val.cpu().numpy(), np.take_along_axis(s, ref, axis=1))
    _lib.set_tuning('topk_slots', None)
    _lib.set_tuning('topk_fast_mode', None)
    _lib.set_tuning('topk_fast_narrow', None)
    _lib.set_tuning('topk_fast_share', None)

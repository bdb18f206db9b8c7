"""Developer: the randomised top-k / SpMM parity sweeps of tests/test_fuzz_gpu.py with other seeds and more cases."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle as O
from igcn_cf_amd import _lib
from igcn_cf_amd.graph import CsrMatrix
from igcn_cf_amd.ops import score_topk, spmm

n_bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    rng = np.random.default_rng(1000 + seed)
    for case in range(80):
        n_users = int(rng.integers(1, 600))
        n_items = int(rng.integers(1, 6000))
        d = int(rng.choice([4, 8, 16, 20, 32, 64, 64, 64, 100, 128, 200, 256]))      # (round 5: 129 ... 256 runs one wave per SIMD)
        k = int(rng.integers(1, min(n_items, 256) + 1))
        if case % 3 == 1:
            _lib.set_tuning('topk_slots', int(rng.integers(1, 8)))
        else:
            _lib.set_tuning('topk_slots', None)
        U = rng.integers(-4, 5, size=(n_users, d)).astype(np.float32)
        I = rng.integers(-4, 5, size=(n_items, d)).astype(np.float32)
        scores = U @ I.T
        kw, ex, ban = {}, None, None
        if case % 2 and n_items > k + 2:
            ex = [sorted(rng.choice(n_items, size=int(rng.integers(0, min(40, n_items - k))), replace=False).tolist()) for _ in range(n_users)]
            rowptr = np.zeros(n_users + 1, dtype=np.int64)
            np.cumsum([len(e) for e in ex], out=rowptr[1:])
            colx = np.array([i for e in ex for i in e], dtype=np.int32)
            kw.update(excl_rowptr=torch.from_numpy(rowptr).cuda(), excl_col=torch.from_numpy(colx).cuda())
        if case % 3 == 0 and n_items > k + 5:
            ban = np.sort(rng.choice(n_items, size=max(1, (n_items - k) // 3), replace=False))
            bm = np.zeros(n_items, dtype=np.uint8); bm[ban] = 1
            kw['banned'] = torch.from_numpy(bm).cuda()
        ids = rng.permutation(n_users).astype(np.int64)
        s = scores[ids].copy()
        if ex is not None:
            for b, u in enumerate(ids):
                if len(ex[u]):
                    s[b, np.asarray(ex[u])] = -np.inf
        if ban is not None:
            s[:, ban] = -np.inf
        ref = O.eval_topk(s, None, None, k=k)
        for prec in ('fp32',):
            idx, val = score_topk(torch.from_numpy(U).cuda(), torch.from_numpy(I).cuda(), k, user_ids=torch.from_numpy(ids).cuda(),
                                  mode='fast' if d in (64, 128) and k <= 60 and case % 4 != 3 else 'exact', **kw)
            ok = np.array_equal(idx.cpu().numpy(), ref) and np.array_equal(val.cpu().numpy(), np.take_along_axis(s, ref, axis=1))
            if not ok:
                n_bad += 1
                print('TOPK MISMATCH', seed, case, n_users, n_items, d, k, prec, flush=True)
    _lib.set_tuning('topk_slots', None)
    for case in range(60):
        n_rows, n_cols = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
        d = int(rng.choice([4, 8, 16, 32, 64, 128, 12, 48]))
        deg = np.minimum((rng.pareto(1.0 + rng.random(), n_rows) * rng.integers(1, 12)).astype(np.int64), n_cols)
        rowptr = np.zeros(n_rows + 1, dtype=np.int64)
        np.cumsum(deg, out=rowptr[1:])
        col = np.concatenate([np.sort(rng.choice(n_cols, size=int(x), replace=False)) for x in deg] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
        val = rng.standard_normal(col.shape[0]).astype(np.float32)
        x = rng.standard_normal((n_cols, d)).astype(np.float32)
        lt = int(rng.choice([64, 128, 256, 1024]))
        csr = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda', long_threshold=lt, segment_len=int(rng.choice([s2 for s2 in (32, 64, 128, 256) if s2 <= lt])))
        y = spmm(csr, torch.from_numpy(x).cuda()).cpu().numpy()
        row = np.repeat(np.arange(n_rows, dtype=np.int64), deg)
        want = O.spmm_coo_f64(row, col.astype(np.int64), val, x, n_rows)
        err = np.abs(y - want).max() / (np.abs(want).max() + 1e-30)
        if err > 1e-4:
            n_bad += 1
            print('SPMM MISMATCH', seed, case, n_rows, n_cols, d, err, flush=True)
# round 3: the two-stage path on float tables with spread norms (order + early exit + bounded fall-back) against the fp32
# sweep, masks included; and the SpMM under random XCD plans against float64
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    rng = np.random.default_rng(5000 + seed)
    for case in range(40):
        n_users, n_items = int(rng.integers(1, 3000)), int(rng.integers(70, 30000))
        k = int(rng.integers(1, 61))
        su, si = float(rng.choice([0., 0.5, 1.5])), float(rng.choice([0., 0.5, 1.5]))
        dd = 128 if case % 3 == 1 else 64
        U = (rng.standard_normal((n_users, dd)) * 0.1 * np.exp(su * rng.standard_normal((n_users, 1)))).astype(np.float32)
        I = (rng.standard_normal((n_items, dd)) * 0.1 * np.exp(si * rng.standard_normal((n_items, 1)))).astype(np.float32)
        if case % 5 == 0:
            I[rng.integers(0, n_items, 5)] = I[0]; U[rng.integers(0, n_users)] = 0.
        kw = {}
        if case % 2:
            ex = [sorted(rng.choice(n_items, size=int(rng.integers(0, 50)), replace=False).tolist()) for _ in range(n_users)]
            rowptr = np.zeros(n_users + 1, dtype=np.int64)
            np.cumsum([len(e) for e in ex], out=rowptr[1:])
            kw.update(excl_rowptr=torch.from_numpy(rowptr).cuda(), excl_col=torch.from_numpy(np.array([i for e in ex for i in e], dtype=np.int32)).cuda())
        if case % 3 == 0:
            bm = (rng.random(n_items) < 0.2).astype(np.uint8)
            kw['banned'] = torch.from_numpy(bm).cuda()
        ids = torch.from_numpy(rng.permutation(n_users).astype(np.int64)).cuda()
        Ut, It = torch.from_numpy(U).cuda(), torch.from_numpy(I).cuda()
        _lib.set_tuning('topk_fast_mode', 2 if case % 4 == 3 else None)
        _lib.set_tuning('topk_fast_narrow', 0 if case % 3 == 2 else None)
        _lib.set_tuning('topk_fast_extra', int(rng.integers(1, 9)) if case % 5 == 4 else None)
        # (round 4: whole sweeps at these sizes — the plan's wave slots tuned down —, so that the warm-up pass runs; its length at random)
        _lib.set_tuning('topk_slots', int(rng.integers(2, 40)) if case % 2 == 0 else None)
        _lib.set_tuning('topk_fast_warm', int(rng.integers(1, 200)) if case % 3 == 0 else None)
        _lib.set_tuning('topk_fast_filter', 0 if case % 7 == 6 else None)
        a = score_topk(Ut, It, k, user_ids=ids, mode='fast', **kw)
        b = score_topk(Ut, It, k, user_ids=ids, mode='exact', **kw)
        if not (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])):
            n_bad += 1
            print('TWO-STAGE MISMATCH', seed, case, n_users, n_items, k, su, si, flush=True)
    for knob in ('topk_fast_mode', 'topk_fast_narrow', 'topk_fast_extra', 'topk_slots', 'topk_fast_warm', 'topk_fast_filter'):
        _lib.set_tuning(knob, None)
    print('seed', seed, 'two-stage cases done, mismatches so far:', n_bad, flush=True)
    for case in range(40):
        nu, ni = int(rng.integers(1, 4000)), int(rng.integers(1, 3000))
        d = int(rng.choice([8, 16, 32, 64, 128]))
        deg = np.minimum((rng.pareto(1.0 + rng.random(), nu) * rng.integers(1, 12)).astype(np.int64) + 1, ni)
        users = np.repeat(np.arange(nu), deg)
        pop = 1. / (np.arange(ni) + 1. + rng.integers(0, 50))
        items = rng.choice(ni, size=users.shape[0], p=pop / pop.sum())
        from igcn_cf_amd.graph import normalized_adjacency_host
        rowptr, col, val = normalized_adjacency_host(np.stack([users, items], 1), nu, ni)
        n = nu + ni
        T = int(rng.choice([4, 16, 64, 112, 300]))
        # (round 5: cut rows added up inside the launch or by the second kernel, the closing segments anywhere in their phase's rows)
        _lib.set_tuning('spmm_fold', int(rng.integers(0, 2)))
        csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n],
                        xcd_plan={'threshold': T, 'assign': str(rng.choice(['affinity', 'spread'])), 'closing_at': float(rng.random()),
                                  'list_order': str(rng.choice(['segments_first', 'rows_first', 'interleaved']))} if case % 4 else None,
                        long_threshold=max(T, 8), segment_len=max(T, 8))
        x = rng.standard_normal((n, d)).astype(np.float32)
        y = spmm(csr, torch.from_numpy(x).cuda()).cpu().numpy()
        row = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
        want = O.spmm_coo_f64(row, col.astype(np.int64), val, x, n)
        err = np.abs(y - want).max() / (np.abs(want).max() + 1e-30)
        if err > 1e-4:
            n_bad += 1
            print('XCD SPMM MISMATCH', seed, case, nu, ni, d, T, err, flush=True)
_lib.set_tuning('spmm_fold', None)
print('done, mismatches:', n_bad)

"""HIP CSR SpMM (through the C ABI) against the CPU oracle.  Tolerance: 1e-4
relative to the output's magnitude (BASELINE.json north_star), fp32."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _random_csr(rng, n_rows, n_cols, degs):
    degs = np.minimum(np.asarray(degs, dtype=np.int64), n_cols)
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(degs, out=rowptr[1:])
    col = np.concatenate([np.sort(rng.choice(n_cols, size=int(k), replace=False)) for k in degs] + [np.zeros(0, dtype=np.int64)])
    val = rng.standard_normal(col.shape[0]).astype(np.float32)
    return rowptr, col.astype(np.int32), val


def _oracle(rowptr, col, val, x, n_rows):
    row = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr))
    return O.spmm_coo_f64(row, col.astype(np.int64), val if val is not None else np.ones(col.shape[0], np.float32), x, n_rows)


@pytest.mark.parametrize('d', [4, 8, 16, 32, 64, 128, 192, 256, 6, 63])
def test_spmm_random_shapes(d):
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    rng = np.random.default_rng(d)
    n_rows, n_cols = 517, 403
    degs = rng.integers(0, 40, size=n_rows)
    degs[:7] = [0, 1, 2, 63, 64, 65, 200]          # empty row, lengths around the 64-wide chunk
    rowptr, col, val = _random_csr(rng, n_rows, n_cols, degs)
    x = rng.standard_normal((n_cols, d)).astype(np.float32)
    csr = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda')
    y = spmm(csr, torch.from_numpy(x).cuda()).cpu().numpy()
    assert _rel_err(y, _oracle(rowptr, col, val, x, n_rows)) < TOL
    assert np.all(y[0] == 0)                        # empty row writes zeros


@pytest.mark.parametrize('d', [64, 128, 32])
def test_spmm_long_rows_segments(d):
    """Rows above the long-row threshold go through the segment + reduce path."""
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    rng = np.random.default_rng(7)
    n_rows, n_cols = 300, 5000
    degs = rng.integers(0, 30, size=n_rows)
    degs[[3, 50, 299]] = [4999, 1025, 2048]
    degs[10] = 1024                                  # exactly at the threshold: ordinary row
    rowptr, col, val = _random_csr(rng, n_rows, n_cols, degs)
    x = rng.standard_normal((n_cols, d)).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    csr = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda', long_threshold=1024, segment_len=512)
    assert csr.n_long == 3 and csr.n_segments == 10 + 3 + 4
    ref = _oracle(rowptr, col, val, x, n_rows)
    y = spmm(csr, xt).cpu().numpy()
    assert _rel_err(y, ref) < TOL
    # a different cut of the same rows gives the same result within rounding, and reruns are bitwise equal
    csr2 = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda', long_threshold=64, segment_len=64)
    y2 = spmm(csr2, xt).cpu().numpy()
    assert _rel_err(y2, ref) < TOL
    assert np.array_equal(spmm(csr2, xt).cpu().numpy(), y2)


def test_spmm_epilogue_and_scales():
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    rng = np.random.default_rng(3)
    n, d = 700, 64
    degs = rng.integers(0, 50, size=n)
    degs[5] = 1500
    rowptr, col, val = _random_csr(rng, n, n, degs)
    x = rng.standard_normal((n, d)).astype(np.float32)
    adds = [rng.standard_normal((n, d)).astype(np.float32) for _ in range(3)]
    rs = rng.random(n).astype(np.float32) + 0.5
    cs = rng.random(n).astype(np.float32) + 0.5
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda')
    y = spmm(csr, torch.from_numpy(x).cuda(), adds=[torch.from_numpy(a).cuda() for a in adds], out_scale=0.25,
             add_scale=0.5, row_scale=torch.from_numpy(rs).cuda(), col_scale=torch.from_numpy(cs).cuda()).cpu().numpy()
    ref = (0.25 * _oracle(rowptr, col, val * cs[col], x, n) + 0.5 * sum(a.astype(np.float64) for a in adds)) * rs[:, None]
    assert _rel_err(y, ref) < TOL
    # val == NULL means all ones
    csr1 = CsrMatrix(rowptr, col, None, (n, n), 'cuda')
    y1 = spmm(csr1, torch.from_numpy(x).cuda()).cpu().numpy()
    assert _rel_err(y1, _oracle(rowptr, col, None, x, n)) < TOL


def test_spmm_rectangular_and_strided():
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    rng = np.random.default_rng(4)
    n_rows, n_cols, d = 333, 1200, 64
    rowptr, col, val = _random_csr(rng, n_rows, n_cols, rng.integers(0, 25, size=n_rows))
    big = rng.standard_normal((n_cols + 10, 2 * d)).astype(np.float32)
    xt = torch.from_numpy(big).cuda()[:, d:]          # leading dimension 2d, offset d
    csr = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda')
    y = spmm(csr, xt).cpu().numpy()
    assert _rel_err(y, _oracle(rowptr, col, val, big[:n_cols, d:], n_rows)) < TOL


def test_spmm_dropout_statistics_and_transpose_consistency():
    """Dropout as a per-edge mask from (seed, edge id): keep-rate ~ 1-p, kept values
    scaled by 1/(1-p) (model.py:263-275), and M / M^T drop the same edges."""
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    rng = np.random.default_rng(5)
    n_rows, n_cols, d = 2000, 1500, 4
    rowptr, col, _ = _random_csr(rng, n_rows, n_cols, rng.integers(5, 60, size=n_rows))
    nnz = col.shape[0]
    p = 0.3
    csr = CsrMatrix(rowptr, col, None, (n_rows, n_cols), 'cuda')
    csr_t = CsrMatrix.transposed(rowptr, col, (n_rows, n_cols), 'cuda')
    ones = torch.ones((n_cols, d), device='cuda')
    y = spmm(csr, ones, keep_prob=1 - p, seed=1234).cpu().numpy()[:, 0]
    kept = y * (1 - p)                                   # number of kept edges per row
    assert np.allclose(kept, np.rint(kept), atol=1e-3)
    rate = kept.sum() / nnz
    assert abs(rate - (1 - p)) < 4 * np.sqrt(p * (1 - p) / nnz)
    # different seed -> different mask; same seed -> identical
    y2 = spmm(csr, ones, keep_prob=1 - p, seed=1235).cpu().numpy()[:, 0]
    assert not np.array_equal(y, y2)
    assert np.array_equal(y, spmm(csr, ones, keep_prob=1 - p, seed=1234).cpu().numpy()[:, 0])
    # <M_d x, z> == <x, M_d^T z> with the same seed
    x = torch.from_numpy(rng.standard_normal((n_cols, d)).astype(np.float32)).cuda()
    z = torch.from_numpy(rng.standard_normal((n_rows, d)).astype(np.float32)).cuda()
    lhs = (spmm(csr, x, keep_prob=1 - p, seed=99).double() * z.double()).sum().item()
    rhs = (x.double() * spmm(csr_t, z, keep_prob=1 - p, seed=99).double()).sum().item()
    assert abs(lhs - rhs) < 1e-4 * max(1., abs(lhs))


def test_propagate_matches_oracle_on_golden_toys(golden):
    from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
    from igcn_cf_amd.ops import propagate_mean
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    adj = O.lightgcn_norm_adj(golden['train_array'], nu, ni)
    rowptr, col, val = normalized_adjacency_host(golden['train_array'], nu, ni)
    np.testing.assert_array_equal(col, adj[1])
    np.testing.assert_array_equal(val, adj[2])           # A_hat values bit-exact with the restated model.py:85-94
    rng = np.random.default_rng(0)
    for d in (64, 128):
        emb = (rng.standard_normal((nu + ni, d)) * 0.1).astype(np.float32)
        csr = CsrMatrix(rowptr, col, val, (nu + ni, nu + ni), 'cuda')
        rep = propagate_mean(csr, torch.from_numpy(emb).cuda(), 3).cpu().numpy()
        assert _rel_err(rep, O.lightgcn_get_rep(adj, emb, 3)) < TOL


@pytest.mark.parametrize('K', [1, 2, 3, 4, 5, 6, 7])
def test_layer_mean_by_the_factored_polynomial_is_the_stack_mean(K):
    """ops.mean_plan evaluates X_0 + A X_0 + ... + A^K X_0 through the factors of 1 + x + ... + x^K (K = 3: (I + A)(I + A^2) X_0,
    one addend read less than stack().mean(), model.py:101-105): against the layer loop in float64, forward and backward, on a
    graph with long rows."""
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
    from igcn_cf_amd.ops import mean_plan, propagate_mean, propagate_mean_backward
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'n_users': 3000, 'n_items': 2000, 'n_inter': 120000, 'seed': 3})
    n = ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, ds.n_users, n], xcd_plan=XCD_PLAN)
    assert len(mean_plan(K)) == K
    g = torch.Generator(device='cuda').manual_seed(K)
    x = torch.randn(n, 64, device='cuda', generator=g) * 0.1
    a = csr.to_torch_coo().double()
    cur, acc = x.double(), x.double().clone()
    for _ in range(K):
        cur = torch.sparse.mm(a, cur)
        acc += cur
    ref = acc / (K + 1)
    # (the third: the reference's own association — every layer kept, all added in the last launch — through the same entry point)
    for got in (propagate_mean(csr, x, K), propagate_mean_backward(csr, x, K), propagate_mean(csr, x, K, plan=mean_plan(K, 'stack'))):
        err = (got.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-30)
        assert float(err.max()) < 1e-6                                                   # (north_star's bound is 1e-4)


def test_propagate_backward_is_adjoint():
    """Linearity + adjointness at full Amazon-book-like size (size-independent
    properties): <P x, z> == <x, P^T z> for P = mean of powers of A_hat."""
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
    from igcn_cf_amd.ops import propagate_mean, propagate_mean_backward
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
    n = ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda')
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(n, 64, device='cuda', generator=g) * 0.1
    z = torch.randn(n, 64, device='cuda', generator=g)
    px = propagate_mean(csr, x, 3)
    ptz = propagate_mean_backward(csr, z, 3)
    lhs, rhs = (px.double() * z.double()).sum().item(), (x.double() * ptz.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * max(1., abs(lhs))
    # linearity
    x2 = torch.randn(n, 64, device='cuda', generator=g) * 0.1
    lin = propagate_mean(csr, 2 * x + x2, 3) - (2 * px + propagate_mean(csr, x2, 3))
    assert lin.abs().max().item() < 1e-5
    # against a float64 torch sparse chain on the device (independent arithmetic)
    a = csr.to_torch_coo().double()
    x64 = x.double()
    l1 = torch.sparse.mm(a, x64); l2 = torch.sparse.mm(a, l1); l3 = torch.sparse.mm(a, l2)
    ref = (x64 + l1 + l2 + l3) / 4
    assert ((px.double() - ref).abs().max() / ref.abs().max()).item() < TOL


def test_c_abi_error_codes_without_launch():
    """Bad arguments are rejected by the C ABI with IGCN_E_* codes before anything is launched
    (include/igcn_hip.h: negative = bad argument, never an exception across the boundary)."""
    import ctypes as C
    from igcn_cf_amd import _lib
    L = _lib.lib()
    x = torch.zeros(8, 64, device='cuda')
    y = torch.zeros(8, 64, device='cuda')
    rowptr = torch.zeros(9, dtype=torch.int64, device='cuda')
    col = torch.zeros(1, dtype=torch.int32, device='cuda')
    nul = (C.c_void_p * 1)()

    def call(rowptr_p, x_p, y_p, d=64, ldx=64, n_adds=0, keep=1.0, n_rows=8):
        return L.igcn_spmm_csr_f32(rowptr_p, col.data_ptr(), None, x_p, ldx, y_p, 64, n_rows, 8, d, 1.0, nul, n_adds, 1.0,
                                   None, None, None, 0, None, 0, None, 256, None, 0, keep, None, 0, 0, None, None, None, None, None, None)
    assert call(None, x.data_ptr(), y.data_ptr()) == -1                       # IGCN_E_NULL
    assert call(rowptr.data_ptr(), x.data_ptr(), y.data_ptr(), d=0) == -2     # IGCN_E_SHAPE
    assert call(rowptr.data_ptr(), x.data_ptr(), y.data_ptr(), d=300) == -2
    assert call(rowptr.data_ptr(), x.data_ptr(), y.data_ptr(), ldx=32) == -2
    assert call(rowptr.data_ptr(), x.data_ptr(), y.data_ptr(), n_adds=9) == -4  # IGCN_E_RANGE
    assert call(rowptr.data_ptr(), x.data_ptr(), y.data_ptr(), keep=0.0) == -4
    assert call(rowptr.data_ptr(), x.data_ptr(), x.data_ptr()) == -4          # in-place
    assert call(rowptr.data_ptr(), x.data_ptr(), y.data_ptr(), n_rows=0) == 0  # empty matrix: nothing to do
    assert call(rowptr.data_ptr(), x.data_ptr(), y.data_ptr()) == 0
    torch.cuda.synchronize()
    # top-k: unsupported shapes are reported by the workspace query and the call
    assert L.igcn_score_topk_workspace_bytes(10, 100, 64, 0) == -1
    assert L.igcn_score_topk_workspace_bytes(10, 100, 66, 5) == -1
    assert L.igcn_score_topk_workspace_bytes(10, 100, 260, 5) == -1                          # d <= 256
    assert L.igcn_score_topk_workspace_bytes(10, 100, 64, 101) == -1
    # ABI v5: xcd_off needs a dealing order; the bounded sweep needs its bounds; the two-stage path its exclusion sizes
    xo = torch.zeros(9, dtype=torch.int64, device='cuda')
    assert L.igcn_spmm_csr_f32(rowptr.data_ptr(), col.data_ptr(), None, x.data_ptr(), 64, y.data_ptr(), 64, 8, 8, 64, 1.0, nul, 0, 1.0,
                               None, None, None, 0, None, 0, None, 256, None, 0, 1.0, None, 0, 0, None, None, None, xo.data_ptr(),
                               None, None) == -1
    # ABI v9: the mask in dealing order comes with the mask itself
    assert L.igcn_spmm_csr_f32(rowptr.data_ptr(), col.data_ptr(), None, x.data_ptr(), 64, y.data_ptr(), 64, 8, 8, 64, 1.0, nul, 0, 1.0,
                               None, None, None, 0, None, 0, None, 256, None, 0, 1.0, None, 0, 0, None, None, None, None,
                               xo.data_ptr(), None) == -1
    items = torch.randn(100, 64, device='cuda')
    oi = torch.empty(8, 5, dtype=torch.int64, device='cuda'); ov = torch.empty(8, 5, device='cuda')
    ws = torch.empty(1 << 22, dtype=torch.uint8, device='cuda')
    assert L.igcn_score_topk_bounded_f32(x.data_ptr(), 64, None, 8, items.data_ptr(), 64, 100, 64, None, None, None, 5, None,
                                         oi.data_ptr(), ov.data_ptr(), ws.data_ptr(), None) == -1
    fl = torch.zeros(9, dtype=torch.int32, device='cuda')
    wsp = (ws.data_ptr() + 255) // 256 * 256
    assert L.igcn_score_topk_fast_workspace_bytes(8, 100, 64, 5, 0, 0) > 0
    assert L.igcn_score_topk_fast_workspace_bytes(8, 100, 32, 5, 0, 0) == -1                  # d = 64 only
    assert L.igcn_score_topk_fast_workspace_bytes(8, 100, 64, 61, 0, 0) == -1                 # k + 4 <= 64
    assert L.igcn_score_topk_fast_f32(x.data_ptr(), 64, None, 8, items.data_ptr(), 64, 100, 64, rowptr.data_ptr(), None, 8, 3, None, 5,
                                      oi.data_ptr(), ov.data_ptr(), fl.data_ptr(), None, wsp, None) == -1   # rowptr without col
    assert L.igcn_score_topk_fast_f32(x.data_ptr(), 64, None, 8, items.data_ptr(), 64, 100, 64, rowptr.data_ptr(), col.data_ptr(), 0, 0,
                                      None, 5, oi.data_ptr(), ov.data_ptr(), fl.data_ptr(), None, wsp, None) == -2  # sizes missing
    assert L.igcn_score_topk_fast_f32(x.data_ptr(), 64, None, 8, items.data_ptr(), 64, 100, 64, None, None, 0, 0, None, 5,
                                      oi.data_ptr(), ov.data_ptr(), None, None, wsp, None) == -1            # no flagged list
    torch.cuda.synchronize()
    from igcn_cf_amd.ops import score_topk
    with pytest.raises(_lib.IgcnError):
        score_topk(x, y, 9)                                                   # k > n_items
    with pytest.raises(_lib.IgcnError):
        score_topk(x, items, 5, mode='fast', lower_bound=torch.zeros(8, device='cuda'))   # a bound goes with the exact sweep
    with pytest.raises(_lib.IgcnError):
        score_topk(x.cpu(), y, 2)                                             # CPU tensor


def test_spmm_row_masks_and_pruned_propagation():
    """Row masks: masked-out rows are left untouched / zeroed, long-row segments included; the pruned
    K-layer pass equals the full one on the needed rows, and so does its backward pass."""
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
    from igcn_cf_amd.ops import PropagateFn, mark_rows, pack_mask_bits, propagate_mean, spmm
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'n_users': 3000, 'n_items': 2000, 'n_inter': 120000, 'zipf_q': 0.})
    n = ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', long_threshold=64, segment_len=64)
    assert csr.n_long > 0
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(n, 64, device='cuda', generator=g) * 0.1
    ids = torch.randint(0, n, (300,), device='cuda', generator=g)
    ids[0] = int(np.argmax(np.diff(rowptr)))                       # a long row is among the needed rows
    m1, m2, b1, b2 = mark_rows(csr, ids)
    assert torch.equal(b2, pack_mask_bits(m2)) and int(b1.view(torch.uint8).sum()) >= 0
    unpacked = ((b2.view(-1, 1) >> torch.arange(32, device='cuda').view(1, -1)) & 1).flatten()[:n].to(torch.uint8)
    assert torch.equal(unpacked, m2)
    deg = np.diff(rowptr)
    want2 = np.zeros(n, dtype=bool)
    for r in ids.cpu().tolist():
        want2[r] = True; want2[col[rowptr[r]:rowptr[r + 1]]] = True
    assert np.array_equal(m2.cpu().numpy().astype(bool), want2) and int(m1.sum()) == len(set(ids.cpu().tolist()))
    full = spmm(csr, x)
    y = torch.full_like(x, 7.0)
    spmm(csr, x, out=y, row_mask=m1, masked_rows_zero=False)
    keep = m1.bool()
    assert torch.equal(y[keep], full[keep]) and torch.all(y[~keep] == 7.0)
    spmm(csr, x, out=y, row_mask=m1, masked_rows_zero=True)
    assert torch.equal(y[keep], full[keep]) and torch.all(y[~keep] == 0.0)
    # column mask: rows of x declared zero are not read (poisoned here), the rest of the sum is unchanged
    xz = x.clone(); xz[~m2.bool()] = 0.0
    want = spmm(csr, xz)
    xp = x.clone(); xp[~m2.bool()] = float('nan')
    assert torch.equal(spmm(csr, xp, col_mask=b2), want)
    assert torch.equal(spmm(csr, xp, col_mask=b2, row_mask=m1, masked_rows_zero=True)[keep], want[keep])
    for K in (1, 2, 3, 4, 5):
        e_full = x.clone().requires_grad_(True)
        e_prun = x.clone().requires_grad_(True)
        r_full = PropagateFn.apply(e_full, csr, csr, K)
        r_prun = PropagateFn.apply(e_prun, csr, csr, K, ids)
        assert torch.equal(r_prun[keep], r_full[keep]) and torch.all(r_prun[~keep] == 0)
        z = torch.zeros_like(x)
        z[keep] = torch.randn(int(keep.sum()), 64, device='cuda', generator=g)
        r_full.backward(z); r_prun.backward(z)
        assert torch.allclose(e_prun.grad, e_full.grad, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize('d', [64, 128, 16])
@pytest.mark.parametrize('plan', ['xcd', 'ordered', 'plain'])
def test_row_mask_in_dealing_order_visits_only_the_wanted_rows(d, plan):
    """igcn_pack_mask_bits_ordered + igcn_spmm_csr_f32(order_bits): the launch walks the set bits of the mask in dealing order instead
    of every entry — same rows computed (bit for bit), every other row untouched, cut rows included; the bit array is the mask
    permuted by row_order (segments: their row's bit) with two zero words behind it."""
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
    from igcn_cf_amd import _lib
    from igcn_cf_amd.ops import RowMarks, mark_rows, spmm
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'n_users': 3000, 'n_items': 2000, 'n_inter': 120000, 'zipf_q': 0., 'seed': 5})
    n = ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
    kw = {'xcd': dict(order_blocks=[0, ds.n_users, n], xcd_plan=XCD_PLAN), 'ordered': dict(order_blocks=[0, ds.n_users, n]), 'plain': {}}[plan]
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', long_threshold=64, segment_len=64, **kw)
    assert csr.n_long > 0 and (csr.row_order is not None) == (plan != 'plain')
    g = torch.Generator(device='cuda').manual_seed(d)
    x = torch.randn(n, d, device='cuda', generator=g) * 0.1
    for n_ids in (1, 40, 700):
        ids = torch.randint(0, n, (n_ids,), device='cuda', generator=g)
        if n_ids == 40:
            ids[0] = int(np.argmax(np.diff(rowptr)))                # a cut row among the wanted ones
        marks = mark_rows(csr, ids)
        assert isinstance(marks, RowMarks) and marks.order_bits_for(csr) is marks.order_bits and marks.order_bits_for(object()) is None
        m1 = marks[0]
        # the bit array, against numpy
        n_virtual = n + csr.n_segments
        order = csr.row_order.cpu().numpy().astype(np.int64) if csr.row_order is not None else np.arange(n_virtual)
        n_order = order.shape[0]                                   # (an XCD plan's lists hold no cut row: fewer entries than n_virtual)
        seg_row = (np.frombuffer(csr.segments.cpu().numpy().tobytes(), dtype=_lib.ROW_SEGMENT_DTYPE)['row'].astype(np.int64)
                   if csr.n_segments else np.zeros(1, dtype=np.int64))
        rows_of = np.where(order < n, order, seg_row[np.maximum(order - n, 0)])
        want_bits = m1.cpu().numpy()[rows_of].astype(bool)
        words = marks.order_bits.cpu().numpy().view(np.uint32)
        assert words.shape[0] == 2 * ((n_virtual + 63) // 64) + 2
        used = 2 * ((n_order + 63) // 64) + 2
        got_bits = ((words[:used, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool).ravel()
        assert np.array_equal(got_bits[:n_order], want_bits) and not got_bits[n_order:].any()
        # the launch
        ref = torch.full((n, d), 7.0, device='cuda')
        spmm(csr, x, out=ref, row_mask=m1, masked_rows_zero=False, adds=[x], out_scale=0.5, add_scale=0.25)
        got = torch.full((n, d), 7.0, device='cuda')
        spmm(csr, x, out=got, row_mask=m1, masked_rows_zero=False, adds=[x], out_scale=0.5, add_scale=0.25, order_bits=marks.order_bits)
        assert torch.equal(got, ref)
        keep = m1.bool()
        assert torch.all(got[~keep] == 7.0) and torch.allclose(got[keep], (0.5 * spmm(csr, x) + 0.25 * x)[keep], rtol=1e-6, atol=1e-7)
        # where masked rows must read as zero the bits are ignored, not misused
        z = torch.full((n, d), 7.0, device='cuda')
        spmm(csr, x, out=z, row_mask=m1, masked_rows_zero=True, order_bits=marks.order_bits)
        assert torch.all(z[~keep] == 0.0) and torch.equal(z[keep], spmm(csr, x)[keep])
    with pytest.raises(Exception):
        spmm(csr, x, order_bits=marks.order_bits)                   # bits without the mask they restate


def test_spmm_matrix_without_entries():
    """A graph with no train pairs: every row is empty, col is an empty array (NULL pointer)."""
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import propagate_mean, spmm
    n = 50
    csr = CsrMatrix(np.zeros(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32), (n, n), 'cuda')
    x = torch.randn(n, 64, device='cuda')
    assert torch.all(spmm(csr, x) == 0)
    assert torch.allclose(propagate_mean(csr, x, 3), x / 4)


def test_device_csr_utilities_match_host_builders():
    """igcn_csr_transpose / igcn_csr_from_sorted_coo against graph.transpose_host and the host rowptr."""
    import ctypes as C
    from igcn_cf_amd import _lib
    from igcn_cf_amd.graph import CsrMatrix, feature_matrix_host, transpose_host
    from igcn_cf_amd.dataset import SyntheticDataset
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'n_users': 3000, 'n_items': 2000, 'n_inter': 90000})
    rowptr, col, row_sum, shape = feature_matrix_host(ds.train_array, ds.n_users, ds.n_items)
    feat = CsrMatrix(rowptr, col, None, shape, 'cuda')
    t = feat.transposed_view()                                        # device path
    h_rowptr, h_col, h_eid = transpose_host(rowptr, col, shape[1])
    np.testing.assert_array_equal(t.rowptr.cpu().numpy(), h_rowptr)
    np.testing.assert_array_equal(t.col.cpu().numpy(), h_col)
    np.testing.assert_array_equal(t.edge_id.cpu().numpy(), h_eid)
    assert t.shape == (shape[1], shape[0]) and t.n_long >= 1          # the two global columns are long rows of F^T
    # rowptr from a sorted COO row array
    rows = torch.from_numpy(np.repeat(np.arange(shape[0], dtype=np.int64), np.diff(rowptr))).cuda()
    out = torch.empty(shape[0] + 1, dtype=torch.int64, device='cuda')
    _lib.check(_lib.lib().igcn_csr_from_sorted_coo(rows.data_ptr(), rows.numel(), shape[0], out.data_ptr(), None), 'from_sorted_coo')
    np.testing.assert_array_equal(out.cpu().numpy(), rowptr)
    # empty matrix
    e = CsrMatrix(np.zeros(11, dtype=np.int64), np.zeros(0, dtype=np.int32), None, (10, 7), 'cuda')
    assert e.transposed_view().shape == (7, 10) and int(e.transposed_view().rowptr.sum()) == 0


def test_launch_shape_does_not_change_results():
    """The nnz hint and the developer grid knob only change how rows are dealt to waves: outputs are
    bit-identical (each row is summed by one wave in storage order whatever the grid is)."""
    import ctypes as C
    from igcn_cf_amd import _lib
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    rng = np.random.default_rng(3)
    n_rows, n_cols, d = 20000, 3000, 64
    degs = np.minimum((rng.pareto(1.2, n_rows) * 5).astype(np.int64), n_cols)
    rowptr, col, val = _random_csr(rng, n_rows, n_cols, degs)
    csr = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda')
    x = torch.randn(n_cols, d, device='cuda')
    ref = spmm(csr, x)
    for bpc in (1, 7, 64, 4096):
        _lib.set_tuning('spmm_blocks_per_cu', bpc)
        assert torch.equal(spmm(csr, x), ref)
    _lib.set_tuning('spmm_blocks_per_cu', None)
    _lib.set_tuning('spmm_multirow', 0)                 # one row per wave: other lane groups, so another summation order
    assert _rel_err(spmm(csr, x).cpu().numpy(), ref.cpu().numpy()) < 1e-5
    _lib.set_tuning('spmm_multirow', None)
    for nnz_hint in (0, 1, 10 ** 12):                                   # unknown / absurdly light / absurdly heavy rows
        y = torch.empty_like(ref)
        nul = (C.c_void_p * 1)()
        rc = _lib.lib().igcn_spmm_csr_f32(csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.val.data_ptr(), x.data_ptr(), d,
                                          y.data_ptr(), d, n_rows, n_cols, d, 1.0, nul, 0, 0.0, None, None,
                                          _lib.ptr(csr.long_rows), csr.n_long, _lib.ptr(csr.segments), csr.n_segments,
                                          _lib.ptr(csr.partial(d)), csr.long_threshold, None, 0, 1.0, None, 0, nnz_hint, None, None, None, None,
                                          None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0 and torch.equal(y, ref)
    # a row order (rows dealt to the waves by descending length inside two blocks) changes who computes a row, not the result
    ordered = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda', order_blocks=[0, n_rows // 3, n_rows])
    assert ordered.n_segments > 0 and sorted(ordered.row_order.cpu().tolist()) == list(range(n_rows + ordered.n_segments))
    assert torch.equal(spmm(ordered, x), ref)


def test_struct_entry_point_the_minimal_binding_fills_eight_fields():
    """igcn_spmm_csr_f32_args (ABI v10): the launch through ONE zero-initialised struct.  What the reference call site has
    (model.py:99-102: a graph, X, the edge values) is the struct's first block — rowptr, col, val, n_rows, n_cols, x, y, d — and
    a binding that fills only those (plus struct_size) gets Y = M X: every optional field reads zero as "off / default".
    Checked against the 34-argument call (same bits), with a SHORTER struct_size (a caller compiled against an older header: the
    library must not read past it), and for the error codes of a struct that is too short / carries unknown flags."""
    import ctypes as C
    from igcn_cf_amd import _lib
    rng = np.random.default_rng(11)
    n_rows, n_cols, d = 3000, 1500, 64
    degs = rng.integers(0, 40, n_rows)                                  # (no row above the cut threshold: no plan needed)
    rowptr, col, val = _random_csr(rng, n_rows, n_cols, degs)
    rp, cl, vl = (torch.from_numpy(a).cuda() for a in (rowptr, col, val))
    x = torch.randn(n_cols, d, device='cuda')
    L = _lib.lib()
    nul = (C.c_void_p * 1)()
    ref = torch.empty(n_rows, d, device='cuda')
    assert L.igcn_spmm_csr_f32(rp.data_ptr(), cl.data_ptr(), vl.data_ptr(), x.data_ptr(), d, ref.data_ptr(), d, n_rows, n_cols, d, 1.0, nul, 0,
                               1.0, None, None, None, 0, None, 0, None, 256, None, 0, 1.0, None, 0, 0, None, None, None, None, None, None) == 0
    want = _oracle(rowptr, col, val, x.cpu().numpy(), n_rows)
    assert _rel_err(ref.cpu().numpy(), want) < TOL
    a = _lib.SpmmArgs()                                                 # ctypes zero-initialises
    a.struct_size = C.sizeof(_lib.SpmmArgs)
    a.rowptr, a.col, a.val, a.n_rows, a.n_cols, a.d = rp.data_ptr(), cl.data_ptr(), vl.data_ptr(), n_rows, n_cols, d
    y = torch.full((n_rows, d), float('nan'), device='cuda')
    a.x, a.y = x.data_ptr(), y.data_ptr()
    assert L.igcn_spmm_csr_f32_args(C.byref(a), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    # a caller compiled against a shorter struct: only the required block (up to and including d)
    prefix = _lib.SpmmArgs.d.offset + 4
    raw = (C.c_uint8 * C.sizeof(_lib.SpmmArgs))()
    C.memmove(raw, C.byref(a), prefix)
    for i in range(prefix, len(raw)):
        raw[i] = 0xFF                                                   # garbage behind what the caller declared: must not be read
    short = C.cast(raw, C.POINTER(_lib.SpmmArgs))
    short.contents.struct_size = prefix
    y.fill_(float('nan'))
    assert L.igcn_spmm_csr_f32_args(short, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    short.contents.struct_size = prefix - 4
    assert L.igcn_spmm_csr_f32_args(short, None) == -2                  # IGCN_E_SHAPE: the required block is not covered
    assert L.igcn_spmm_csr_f32_args(None, None) == -1
    a.flags = 4
    assert L.igcn_spmm_csr_f32_args(C.byref(a), None) == -4             # unknown flag bit
    a.flags = 0
    a.n_adds = 9
    assert L.igcn_spmm_csr_f32_args(C.byref(a), None) == -4
    a.n_adds = 1                                                        # an addend announced, none given
    assert L.igcn_spmm_csr_f32_args(C.byref(a), None) == -1
    a.n_adds = 0
    a.tune_fold = -1
    assert L.igcn_spmm_csr_f32_args(C.byref(a), None) == -4
    a.tune_fold = 0
    a.y = x.data_ptr()
    assert L.igcn_spmm_csr_f32_args(C.byref(a), None) == -4             # in place
    # the struct's layout is what the header says: 8-byte aligned pointers, no hidden padding in the required block
    assert _lib.SpmmArgs.rowptr.offset == 8 and _lib.SpmmArgs.d.offset == 64 and C.sizeof(_lib.SpmmArgs) % 8 == 0


def test_a_plain_c_program_drives_the_library(tmp_path):
    """tests/c_caller/spmm_caller.c: a C program — no Python, no torch in its process — that includes include/igcn_hip.h, links
    libigcn_hip.so, fills the eight required fields of a zeroed igcn_spmm_args and compares Y = M X with a host loop, then the
    positional form (same bits) and the refusals.  The boundary as a cgo / JNI / plain C binding would use it."""
    import os
    import shutil
    import subprocess
    from igcn_cf_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    if shutil.which('gcc') is None or not os.path.exists(os.path.join(rocm, 'include', 'hip', 'hip_runtime_api.h')):
        pytest.skip('no gcc / HIP runtime headers on this box')
    exe = str(tmp_path / 'spmm_caller')
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    # gcc, C99: the header and the caller are plain C; the HIP runtime is linked for hipMalloc / hipMemcpy only
    cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(root, 'include'), '-I', os.path.join(rocm, 'include'),
           os.path.join(root, 'tests', 'c_caller', 'spmm_caller.c'), '-L', lib_dir, '-l:libigcn_hip.so', '-L', os.path.join(rocm, 'lib'),
           '-lamdhip64', '-lm', '-Wl,-rpath,' + lib_dir, '-Wl,-rpath,' + os.path.join(rocm, 'lib'), '-o', exe]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert b.returncode == 0, b.stdout.decode()[-3000:]
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = r.stdout.decode()
    assert r.returncode == 0 and out.strip().endswith('ok'), out[-2000:]


def test_per_call_launch_knobs_keep_concurrent_streams_apart():
    """igcn_set_tuning is process-wide; the SpMM's three result-neutral knobs can instead ride in the call (igcn_spmm_args.tune_* /
    ops.spmm(tune=...)).  Launches on several streams and threads at once, each with its own grid / rows-per-wave / fold choice and
    none touching the process-wide switch: every result equals the default launch's — bit for bit where only the grid differs, to
    rounding where the lanes that add a row up differ — and the process-wide values are what they were."""
    import threading
    from igcn_cf_amd import _lib
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    rng = np.random.default_rng(5)
    n_rows, n_cols, d = 30000, 4000, 64
    degs = np.minimum((rng.pareto(1.2, n_rows) * 5).astype(np.int64), n_cols)
    rowptr, col, val = _random_csr(rng, n_rows, n_cols, degs)
    x = torch.randn(n_cols, d, device='cuda')
    base = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda', order_blocks=[0, n_rows])
    assert base.n_segments > 0
    ref = spmm(base, x)
    torch.cuda.synchronize()
    knobs = [{'blocks_per_cu': 1}, {'blocks_per_cu': 64}, {'multirow': 0}, {'fold': 1}, {'blocks_per_cu': 7, 'fold': 1}, None]
    results, errors = [None] * len(knobs), []

    def body(i):
        try:
            csr = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda', order_blocks=[0, n_rows])      # (its own partial-sum workspace)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                xs = x.clone()
                for _ in range(20):
                    y = spmm(csr, xs, tune=knobs[i])
            st.synchronize()
            results[i] = y
        except BaseException as e:                                       # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=body, args=(i,)) for i in range(len(knobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for i, k in enumerate(knobs):
        if k and 'multirow' in k:
            assert _rel_err(results[i].cpu().numpy(), ref.cpu().numpy()) < 1e-5, k
        else:
            assert torch.equal(results[i], ref), k
    assert torch.equal(spmm(base, x), ref)                               # the process-wide defaults were never touched
    # a per-call knob wins over the process-wide one, and only for its call
    _lib.set_tuning('spmm_blocks_per_cu', 3)
    try:
        assert torch.equal(spmm(base, x, tune={'blocks_per_cu': 4096}), ref) and torch.equal(spmm(base, x), ref)
    finally:
        _lib.set_tuning('spmm_blocks_per_cu', None)


@pytest.mark.parametrize('d', [64, 128, 16])
def test_cut_rows_added_up_inside_the_launch_give_the_two_launch_bits(d):
    """igcn_set_tuning("spmm_fold", 1) (opt-in, include/igcn_hip.h: closing segments): every segment of a cut row counts itself in
    on the row's arrival counter, the row's closing segment — dealt later in its list — polls the count, adds the partial sums up
    in slot order and applies the epilogue, all inside the one launch.  Against the default two-launch form (the same slot order
    in spmm_long_rows_reduce_kernel): BIT-EQUAL — plain long-row plan and XCD plan, the layer-mean epilogue, a row scale, row masks
    with and without zeroing of the rows skipped, dropout, and launch after launch (the counters go back to zero)."""
    from igcn_cf_amd import _lib
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    rng = np.random.default_rng(11)
    nu, ni = 3000, 2000
    n = nu + ni
    deg_u = rng.integers(5, 40, nu)
    deg_i = np.minimum((rng.pareto(0.9, ni) * 30).astype(np.int64) + 1, nu)          # power-law item rows: many cut rows
    rows_u = [np.sort(rng.choice(ni, size=k, replace=False)) + nu for k in deg_u]
    rows_i = [np.sort(rng.choice(nu, size=k, replace=False)) for k in deg_i]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.concatenate([deg_u, deg_i]), out=rowptr[1:])
    col = np.concatenate(rows_u + rows_i).astype(np.int32)
    val = rng.random(col.shape[0]).astype(np.float32)
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(n, d, device='cuda', generator=g)
    adds = [torch.randn(n, d, device='cuda', generator=g) for _ in range(3)]
    scale = torch.rand(n, device='cuda', generator=g) + 0.5
    mask = (torch.rand(n, device='cuda', generator=g) < 0.4).to(torch.uint8)
    try:
        for plan in (None, {'threshold': 48}):
            csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], long_threshold=64, segment_len=64, xcd_plan=plan)
            assert csr.n_long > 50 and csr.closing_segments
            sg = csr.segments.view(torch.int32).view(-1, 6)[:, 5]
            assert int((sg < 0).sum()) == csr.n_long                                  # one closing segment per cut row
            cases = [dict(), dict(adds=adds, out_scale=0.25, add_scale=0.25), dict(row_scale=scale),
                     dict(row_mask=mask, masked_rows_zero=True), dict(row_mask=mask, masked_rows_zero=False),
                     dict(keep_prob=0.7, seed=99)]
            for kw in cases:
                outs = []
                for on in (0, 1, 1):                                                  # the second folded launch finds its counters at zero
                    _lib.set_tuning('spmm_fold', on)
                    y = torch.full((n, d), 7.0, device='cuda')
                    spmm(csr, x, out=y, **kw)
                    outs.append(y)
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (plan, sorted(kw))
            ref = torch.sparse.mm(csr.to_torch_coo().double(), x.double())
            _lib.set_tuning('spmm_fold', 1)
            assert _rel_err(spmm(csr, x).cpu().numpy(), ref.cpu().numpy()) < 1e-5
    finally:
        _lib.set_tuning('spmm_fold', None)


def test_config5_bipartite_graph_all_eight_rank_shares():
    """BASELINE config 5 AS WRITTEN: a bipartite 10 M users x 2 M items x ~500 M edges graph generated in HBM with the
    SURVEY 8(d) rules (log-normal user degrees >= 7, Zipf-Mandelbrot items over a random permutation, de-duplicated),
    cut with ShardLayout.balanced(world = 8), every rank's share — its user block (gathering from the 2 M item rows)
    and its item block (gathering from the 10 M user rows) — run one after the other on this one GPU against the full
    replicated 12 M x 128 operand (6.1 GB; the 'halves' exchange of dist.py is a device copy here, SURVEY 8(e) caveat):
    256 sampled rows per rank against float64 <= 1e-4, sum of the local nonzeros == nnz(A_hat), shares nnz-balanced,
    and the adjoint identity on one share's transposed view."""
    from igcn_cf_amd.dist import ShardLayout
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    from igcn_cf_amd.synth import BipartiteGraphDevice, check_rows_f64
    d, world = 128, 8
    g = BipartiteGraphDevice(10_000_000, 2_000_000, 500_000_000, 'cuda', seed=2021)
    assert abs(g.n_edges / 5e8 - 1) < 0.03 and int(g.deg_u.min()) >= 5
    layout = ShardLayout.balanced(g.rowptr_host(), g.n_users, g.n_items, world)
    gen = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(g.n, d, device='cuda', generator=gen) * 0.1
    total = 0
    for r in range(world):
        csr, grow = g.rank_share(layout, r)
        total += csr.nnz
        assert abs(csr.nnz / (g.nnz / world) - 1) < 0.02, (r, csr.nnz)                      # balanced by nonzeros
        assert csr.n_long > 0                                                               # popular items: the long-row pass too
        y = spmm(csr, x)
        rows = torch.randint(0, csr.shape[0], (256,), device='cuda', generator=gen).tolist()
        assert check_rows_f64(csr, x, y, rows) < 1e-4, r
        if r == 3:                                                                           # <Y, A X> = <A^T Y, X>
            ones = CsrMatrix.from_device(csr.rowptr, csr.col, None, csr.shape)
            yt = torch.randn(csr.shape[0], d, device='cuda', generator=gen)
            lhs = (spmm(ones, x).double() * yt.double()).sum().item()
            rhs = (spmm(ones.transposed_view(), yt).double() * x.double()).sum().item()
            assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
            del ones, yt
        del csr, grow, y
    assert total == g.nnz


def test_device_graph_builders_are_bit_identical_to_the_host_ones(golden):
    """normalized_adjacency_device / feature_matrix_device (what the models build their graphs with, in HBM) against
    the host builders that are pinned to the reference-run fixtures: same structure, bit-identical values, on the golden
    toys (duplicate pairs, empty users, partial template maps) and on a 5000 x 4000 synthetic split."""
    from igcn_cf_amd.dataset import ProcessedDataset, SyntheticDataset
    from igcn_cf_amd.graph import (feature_matrix_device, feature_matrix_host, normalized_adjacency_device,
                                   normalized_adjacency_host)
    cases = [ProcessedDataset({'name': 'ProcessedDataset', 'path': golden['path'], 'device': 'cuda'}),
             SyntheticDataset({'name': 'SyntheticDataset', 'n_users': 5000, 'n_items': 4000, 'n_inter': 150000, 'zipf_q': 0.})]
    for ds in cases:
        nu, ni = ds.n_users, ds.n_items
        rp, col, val = normalized_adjacency_host(ds.train_array, nu, ni)
        a = normalized_adjacency_device(ds.train_array, nu, ni, 'cuda')
        np.testing.assert_array_equal(a.rowptr.cpu().numpy(), rp)
        np.testing.assert_array_equal(a.col.cpu().numpy(), col)
        np.testing.assert_array_equal(a.val.cpu().numpy(), val)                 # bit-exact, not allclose
        half_u = {int(u): j for j, u in enumerate(range(0, nu, 2))}
        third_i = {int(i): j for j, i in enumerate(range(1, ni, 3))}
        for um, im in ((None, None), (half_u, third_i)):
            frp, fcol, frow_sum, shape = feature_matrix_host(ds.train_array, nu, ni, um, im)
            f, row_sum = feature_matrix_device(ds.train_array, nu, ni, um, im, 'cuda')
            assert f.shape == shape and f.val is None
            np.testing.assert_array_equal(f.rowptr.cpu().numpy(), frp)
            np.testing.assert_array_equal(f.col.cpu().numpy(), fcol)
            np.testing.assert_array_equal(row_sum.cpu().numpy(), frow_sum)


def test_dropout_seed_from_device_memory_matches_the_launch_argument():
    """seed_dev (ABI v4): the same 64-bit seed read from device memory drops the same edges as when it is passed as a
    launch argument — on the matrix and on its transposed view (edge ids) — and a new value written there changes
    the mask without any new binding (what a captured HIP graph replays)."""
    from igcn_cf_amd import _lib
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import spmm
    rng = np.random.default_rng(5)
    n_rows, n_cols, d = 700, 500, 64
    dense = (rng.random((n_rows, n_cols)) < 0.05)
    row, col = np.nonzero(dense)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(row, minlength=n_rows))]).astype(np.int64)
    csr = CsrMatrix(rowptr, col.astype(np.int32), None, (n_rows, n_cols), 'cuda')
    x = torch.randn(n_cols, d, device='cuda')
    g = torch.randn(n_rows, d, device='cuda')
    csr_t = csr.transposed_view()
    seed_dev = torch.zeros(1, dtype=torch.int64, device='cuda')
    outs = []
    for seed in (123456789012345, 987654321):
        seed_dev.fill_(seed)
        a = spmm(csr, x, keep_prob=0.6, seed=seed)
        b = spmm(csr, x, keep_prob=0.6, seed=seed_dev)
        assert torch.equal(a, b)
        assert torch.equal(spmm(csr_t, g, keep_prob=0.6, seed=seed), spmm(csr_t, g, keep_prob=0.6, seed=seed_dev))
        outs.append(b)
    assert not torch.equal(outs[0], outs[1])
    with pytest.raises(_lib.IgcnError):
        spmm(csr, x, keep_prob=0.6, seed=torch.zeros(1, dtype=torch.int32, device='cuda'))


@pytest.mark.parametrize('d', [64, 6])
def test_col_mask_never_reads_masked_source_rows(d):
    """col_mask: a source row whose bit is clear is NOT read — it may hold uninitialised memory (the first backward
    hop of a training step leaves rows outside the batch's neighbourhood unwritten).  NaNs in the masked rows must
    not reach the result, on the vector path (d = 64) and on the scalar path for odd shapes (d = 6)."""
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import pack_mask_bits, spmm
    rng = np.random.default_rng(3)
    n_rows, n_cols = 200, 150
    rowptr, col, val = _random_csr(rng, n_rows, n_cols, rng.integers(0, 30, size=n_rows))
    x = rng.standard_normal((n_cols, d)).astype(np.float32)
    keep = rng.random(n_cols) < 0.5
    x_clean = x * keep[:, None]
    x_dirty = x.copy()
    x_dirty[~keep] = np.nan
    csr = CsrMatrix(rowptr, col, val, (n_rows, n_cols), 'cuda')
    bits = pack_mask_bits(torch.from_numpy(keep.astype(np.uint8)).cuda())
    y = spmm(csr, torch.from_numpy(x_dirty).cuda(), col_mask=bits).cpu().numpy()
    assert np.isfinite(y).all()
    assert _rel_err(y, _oracle(rowptr, col, val, x_clean, n_rows)) < TOL


@pytest.mark.parametrize('d', [64, 128, 32, 8])
def test_xcd_plan_changes_who_computes_what_not_the_result(d):
    """igcn_spmm_csr_f32 with xcd_off (ABI v5): eight per-XCD lists, rows above the threshold cut at operand-slice
    boundaries.  Against the float64 oracle <= 1e-4; against the plain plan <= 1e-5 (a cut row is summed piecewise);
    bit-identical from launch to launch and for every grid size; row masks / zero-filled masked rows / col masks / the
    fused epilogue behave as without the plan; a matrix without entries and one without rows to cut are fine."""
    from igcn_cf_amd import _lib
    from igcn_cf_amd.graph import CsrMatrix
    from igcn_cf_amd.ops import pack_mask_bits, spmm
    rng = np.random.default_rng(5)
    nu, ni = 9000, 4000
    n = nu + ni
    deg_u = np.minimum((rng.pareto(1.3, nu) * 6).astype(np.int64) + 1, 600)
    users = np.repeat(np.arange(nu), deg_u)
    pop = 1. / (np.arange(ni) + 20.)
    items = rng.choice(ni, size=users.shape[0], p=pop / pop.sum())
    from igcn_cf_amd.graph import normalized_adjacency_host
    rowptr, col, val = normalized_adjacency_host(np.stack([users, items], 1), nu, ni)
    x = (rng.standard_normal((n, d)) * 0.1).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    plain = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n])
    ref64 = _oracle(rowptr, col, val, x, n)
    y_plain = spmm(plain, xt)
    for T in (16, 112):
        plan = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan={'threshold': T})
        assert plan.xcd_off is not None and plan.n_long > 0 and plan.long_threshold == T
        y = spmm(plan, xt)
        assert _rel_err(y.cpu().numpy(), ref64) < TOL
        assert _rel_err(y.cpu().numpy(), y_plain.cpu().numpy()) < 1e-5
        assert torch.equal(spmm(plan, xt), y)
        for bpc in (1, 3, 64):
            _lib.set_tuning('spmm_blocks_per_cu', bpc)
            assert torch.equal(spmm(plan, xt), y)
        _lib.set_tuning('spmm_blocks_per_cu', None)
        # epilogue + row mask (+ zero fill) + col mask
        mask = torch.from_numpy((rng.random(n) < 0.3).astype(np.uint8)).cuda()
        keep = torch.from_numpy((rng.random(n) < 0.6).astype(np.uint8)).cuda()
        add = torch.randn(n, d, device='cuda')
        rs = torch.rand(n, device='cuda') + 0.5
        kw = dict(adds=[add, xt], out_scale=0.25, add_scale=0.5, row_scale=rs, row_mask=mask, masked_rows_zero=True,
                  col_mask=pack_mask_bits(keep))
        a, b = spmm(plan, xt, **kw), spmm(plain, xt, **kw)
        assert _rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
        assert float(a[mask == 0].abs().max()) == 0.0
        out = torch.full((n, d), 7., device='cuda')
        spmm(plan, xt, out=out, row_mask=mask, masked_rows_zero=False)
        assert bool((out[mask == 0] == 7.).all()) and _rel_err(out[mask == 1].cpu().numpy(), y[mask == 1].cpu().numpy()) == 0.0
    none_cut = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan={'threshold': 100000})
    assert none_cut.n_long == 0 and none_cut.xcd_off is not None
    assert _rel_err(spmm(none_cut, xt).cpu().numpy(), ref64) < TOL
    empty = CsrMatrix(np.zeros(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32), (n, n), 'cuda',
                      order_blocks=[0, nu, n], xcd_plan={'threshold': 16})
    assert float(spmm(empty, xt).abs().max()) == 0.0


def test_default_graph_builder_uses_the_xcd_plan_and_matches_the_plain_one(golden):
    """LightGCN.generate_graph builds A_hat with graph.XCD_PLAN (built in HBM by the same torch code as on the host);
    the propagation through it equals the one through the plain plan and the oracle's."""
    from igcn_cf_amd.graph import XCD_PLAN, normalized_adjacency_device
    from igcn_cf_amd.ops import propagate_mean
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    a = normalized_adjacency_device(golden['train_array'], nu, ni, 'cuda')
    b = normalized_adjacency_device(golden['train_array'], nu, ni, 'cuda', xcd_plan=None)
    assert a.xcd_off is not None and a.long_threshold == XCD_PLAN['threshold'] and b.xcd_off is None
    assert torch.equal(a.col, b.col) and torch.equal(a.val, b.val)
    x = torch.randn(nu + ni, 64, device='cuda')
    ya, yb = propagate_mean(a, x, 3), propagate_mean(b, x, 3)
    assert _rel_err(ya.cpu().numpy(), yb.cpu().numpy()) < 1e-5
    adj = O.lightgcn_norm_adj(golden['train_array'], nu, ni)
    assert _rel_err(ya.cpu().numpy(), O.lightgcn_get_rep(adj, x.cpu().numpy(), 3)) < TOL

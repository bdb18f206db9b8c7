"""Row-sharded propagation (igcn_cf_amd/dist.py) under gloo, world_size 2, on CPU.
The local product is injected (a checker implementation over torch CPU tensors) so the
partitioning, padded layout, alternating half-step order and in-place all-gathers are
exercised without a GPU; the result must equal the unsharded oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O


class CpuCsr:
    def __init__(self, rowptr, col, val, shape):
        self.rowptr, self.col, self.val, self.shape = rowptr, col, val, shape


def cpu_spmm(csr, x, out=None, adds=(), out_scale=1.0, add_scale=1.0):
    row = np.repeat(np.arange(csr.shape[0], dtype=np.int64), np.diff(csr.rowptr))
    y = O.spmm_coo(row, csr.col.astype(np.int64), csr.val, x.numpy(), n_rows=csr.shape[0]) * np.float32(out_scale)
    for a in adds:
        y += np.float32(add_scale) * a.numpy()
    out.copy_(torch.from_numpy(y))
    return out


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, train_array, nu, ni, emb, n_layers, exchange, balance, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from igcn_cf_amd.dist import RowShardedPropagator
        prop = RowShardedPropagator(train_array, nu, ni, n_layers, rank, world, 'cpu', spmm_fn=cpu_spmm,
                                    csr_factory=lambda rp, c, v, shape: CpuCsr(rp, c, v, shape), exchange=exchange,
                                    balance=balance)
        L = prop.layout
        (ulo, uhi), (ilo, ihi) = L.user_rows(rank), L.item_rows(rank)
        prop.load_local_embedding(torch.from_numpy(emb[ulo:uhi]), torch.from_numpy(emb[nu + ilo:nu + ihi]))
        ru, ri = prop.propagate()
        full = prop.gather_full_rep(ru, ri)
        ret[rank] = (ru[:uhi - ulo].numpy().copy(), ri[:ihi - ilo].numpy().copy(), full.numpy().copy(),
                     prop.local_nnz, prop.global_nnz, (ulo, uhi, ilo, ihi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_layers,exchange,balance', [(1, 'halves', True), (2, 'halves', False), (3, 'halves', True),
                                                       (4, 'halves', True), (1, 'fused', True), (3, 'fused', True),
                                                       (4, 'fused', False)])
def test_row_sharded_propagation_equals_unsharded(golden, n_layers, exchange, balance):
    """Both exchanges (one all-gather per layer on the interleaved layout / two overlapped half-layer gathers),
    equal-row and nnz-balanced (unequal, padded) blocks, 1-4 layers: owned rows and the gathered table equal the
    unsharded oracle."""
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    ta = golden['train_array']
    rng = np.random.default_rng(0)
    emb = (rng.standard_normal((nu + ni, 16)) * 0.1).astype(np.float32)
    ref = O.lightgcn_get_rep(O.lightgcn_norm_adj(ta, nu, ni), emb, n_layers)
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ta, nu, ni, emb, n_layers, exchange, balance, ret), nprocs=world, join=True)
    nnz, users_seen, items_seen = 0, 0, 0
    for r in range(world):
        ru, ri, full, lnnz, gnnz, (ulo, uhi, ilo, ihi) = ret[r]
        users_seen += uhi - ulo
        items_seen += ihi - ilo
        np.testing.assert_allclose(ru, ref[ulo:uhi], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ri, ref[nu + ilo:nu + ihi], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(full, ref, rtol=1e-5, atol=1e-7)
        nnz += lnnz
    assert nnz == gnnz and users_seen == nu and items_seen == ni          # every edge / node owned exactly once


def test_shard_layout_padding_and_balance():
    from igcn_cf_amd.dist import ShardLayout, local_blocks_host
    L = ShardLayout(5, 3, 4)                              # equal-row blocks of 2 users / 1 item, ranks with empty tails
    assert (L.bu, L.bi, L.pu, L.pi) == (2, 1, 8, 4)
    assert L.user_rows(2) == (4, 5) and L.user_rows(3) == (5, 5) and L.item_rows(3) == (3, 3)
    np.testing.assert_array_equal(L.pad_index([0, 4, 5, 7]), [0, 4, 8, 10])
    rowptr = np.arange(9, dtype=np.int64)                 # one entry per row
    col = np.array([5, 6, 7, 5, 6, 0, 1, 2], dtype=np.int32)
    val = np.ones(8, dtype=np.float32)
    (urp, ucol, _), (irp, icol, _) = local_blocks_host(rowptr, col, val, L, 3)
    assert urp.tolist() == [0, 0, 0] and irp.tolist() == [0, 0]          # rank 3 owns padding only
    (urp, ucol, _), (irp, icol, _) = local_blocks_host(rowptr, col, val, L, 2)
    assert urp.tolist() == [0, 1, 1] and ucol.tolist() == [9] and irp.tolist() == [0, 1] and icol.tolist() == [2]
    # interleaved layout of the fused exchange: [rank 0: 2 users, 1 item | rank 1: ... ]
    F = ShardLayout(5, 3, 4, fused=True)
    np.testing.assert_array_equal(F.pad_index([0, 1, 2, 4, 5, 6, 7]), [0, 1, 3, 6, 2, 5, 8])
    # nnz-balanced blocks on a power-law graph: row counts differ, nonzeros per block nearly equal
    rng = np.random.default_rng(0)
    nu, ni, P = 4000, 3000, 8
    deg_u = rng.integers(5, 30, nu)
    deg_i = np.maximum((rng.pareto(1.1, ni) * 8).astype(np.int64), 1)
    rowptr = np.concatenate([[0], np.cumsum(np.concatenate([deg_u, deg_i]))]).astype(np.int64)
    B = ShardLayout.balanced(rowptr, nu, ni, P)
    E = ShardLayout(nu, ni, P)
    def spread(L, lo_of, base):
        nnz = [rowptr[base + lo_of(r)[1]] - rowptr[base + lo_of(r)[0]] for r in range(P)]
        return max(nnz) / (sum(nnz) / P)
    assert spread(B, B.item_rows, nu) < spread(E, E.item_rows, nu) and spread(B, B.item_rows, nu) < 1.6       # one huge row can outweigh a block
    assert spread(B, B.user_rows, 0) < 1.02
    assert len({B.item_rows(r)[1] - B.item_rows(r)[0] for r in range(P)}) > 1       # unequal row counts
    rank, local = B.owner(np.arange(nu + ni))
    for r in range(P):                                                               # owner() inverts the block maps
        (ulo, uhi), (ilo, ihi) = B.user_rows(r), B.item_rows(r)
        assert np.all(rank[ulo:uhi] == r) and np.all(rank[nu + ilo:nu + ihi] == r)
        np.testing.assert_array_equal(local[ulo:uhi], np.arange(uhi - ulo))
        np.testing.assert_array_equal(local[nu + ilo:nu + ihi], B.bu + np.arange(ihi - ilo))


# ---- sharded training step (forward + backward + Adam) vs the same step unsharded ----------------
def torch_bpr_terms(rep, emb, users, pos, neg, n_users):
    """trainer.py:238-243 + model.py:110-116 in torch (checker implementation for the CPU test)."""
    ur, pr, nr = rep[users], rep[n_users + pos], rep[n_users + neg]
    ue, pe, ne = emb[users], emb[n_users + pos], emb[n_users + neg]
    bpr = torch.nn.functional.softplus((ur * nr).sum(1) - (ur * pr).sum(1)).mean()
    l2 = ((ue ** 2).sum(1) + (pe ** 2).sum(1) + (ne ** 2).sum(1)).mean()
    return torch.stack([bpr, l2])


def torch_bpr_terms_rows(rows_rep, rows_emb, B):
    """The same on the exchanged batch rows [users | positives | negatives] (dist.ShardedLightGCN's loss_fn)."""
    ur, pr, nr = rows_rep[:B], rows_rep[B:2 * B], rows_rep[2 * B:]
    bpr = torch.nn.functional.softplus((ur * nr).sum(1) - (ur * pr).sum(1)).mean()
    return torch.stack([bpr, (rows_emb ** 2).sum(1).view(3, B).sum(0).mean()])


def _train_worker(rank, world, port, path, emb, batch, n_layers, exchange, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from igcn_cf_amd.dataset import ProcessedDataset
        from igcn_cf_amd.dist import ShardedLightGCN
        ds = ProcessedDataset({'name': 'ProcessedDataset', 'path': path, 'device': 'cpu'})
        model = ShardedLightGCN(ds, emb.shape[1], n_layers, rank, world, 'cpu', spmm_fn=cpu_spmm,
                                csr_factory=lambda rp, c, v, shape: CpuCsr(rp, c, v, shape), loss_fn=torch_bpr_terms_rows,
                                full_embedding=torch.from_numpy(emb), exchange=exchange)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        b = torch.from_numpy(batch)
        losses = []
        for _ in range(2):
            terms = model.bpr_loss_terms(b[:, 0], b[:, 1], b[:, 2])
            loss = terms[0] + 1e-2 * terms[1]
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(float(loss))
        ret[rank] = (losses, model.full_embedding().numpy().copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('exchange', ['fused', 'halves'])
def test_sharded_training_step_equals_unsharded(golden, exchange):
    """Two Adam steps of the row-sharded model (nnz-balanced unequal blocks, batch rows exchanged by one
    all-reduce, duplicate ids in the batch) against the same steps on the dense unsharded chain."""
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    ta = golden['train_array']
    n_layers = 3
    rng = np.random.default_rng(1)
    emb = (rng.standard_normal((nu + ni, 8)) * 0.1).astype(np.float32)
    batch = np.stack([rng.integers(0, nu, 64), rng.integers(0, ni, 64), rng.integers(0, ni, 64)], axis=1).astype(np.int64)
    # unsharded reference: dense torch chain of the oracle's normalised adjacency
    row, col, val = O.lightgcn_norm_adj(ta, nu, ni)
    a = torch.sparse_coo_tensor(np.stack([row, col]), val, (nu + ni, nu + ni)).to_dense()
    e = torch.nn.Parameter(torch.from_numpy(emb.copy()))
    opt = torch.optim.Adam([e], lr=1e-2)
    b = torch.from_numpy(batch)
    ref_losses = []
    for _ in range(2):
        x, layers = e, [e]
        for _l in range(n_layers):
            x = a @ x
            layers.append(x)
        rep = torch.stack(layers).mean(0)
        terms = torch_bpr_terms(rep, e, b[:, 0], b[:, 1], b[:, 2], nu)
        loss = terms[0] + 1e-2 * terms[1]
        opt.zero_grad(); loss.backward(); opt.step()
        ref_losses.append(float(loss))
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_train_worker, args=(world, _free_port(), golden['path'], emb, batch, n_layers, exchange, ret), nprocs=world, join=True)
    for r in range(world):
        losses, full = ret[r]
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-5)
        np.testing.assert_allclose(full, e.detach().numpy(), rtol=1e-4, atol=1e-6)


# ---- embedding-column sharding ---------------------------------------------------------------------
def _col_worker(rank, world, port, path, emb, batch, n_layers, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from igcn_cf_amd.dataset import ProcessedDataset
        from igcn_cf_amd.dist import ColumnShardedLightGCN
        ds = ProcessedDataset({'name': 'ProcessedDataset', 'path': path, 'device': 'cpu'})
        nu, ni = ds.n_users, ds.n_items
        row, col, val = O.lightgcn_norm_adj(ds.train_array, nu, ni)
        a = torch.sparse_coo_tensor(np.stack([row, col]), val, (nu + ni, nu + ni)).to_dense()

        def propagate(e):                                   # checker implementation of the local K-layer pass
            x, layers = e, [e]
            for _ in range(n_layers):
                x = a @ x
                layers.append(x)
            return torch.stack(layers).mean(0)

        def loss(rep, e, users, pos, neg, n_users, reduce_fn):   # partial dots -> all-reduce -> softplus
            ur, pr, nr = rep[users], rep[n_users + pos], rep[n_users + neg]
            ue, pe, ne = e[users], e[n_users + pos], e[n_users + neg]
            dots = torch.stack([(ur * pr).sum(1), (ur * nr).sum(1), (ue ** 2).sum(1) + (pe ** 2).sum(1) + (ne ** 2).sum(1)])
            total = dots.detach().clone()
            reduce_fn(total)
            full = dots + (total - dots.detach())           # value of the sum, gradient of the local part
            return torch.stack([torch.nn.functional.softplus(full[1] - full[0]).mean(), full[2].mean()])

        model = ColumnShardedLightGCN(ds, emb.shape[1], n_layers, rank, world, 'cpu', full_embedding=torch.from_numpy(emb),
                                      propagate_fn=propagate, loss_fn=loss)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        b = torch.from_numpy(batch)
        losses = []
        for _ in range(2):
            terms = model.bpr_loss_terms(b[:, 0], b[:, 1], b[:, 2])
            l = terms[0] + 1e-2 * terms[1]
            opt.zero_grad(); l.backward(); opt.step()
            losses.append(float(l))
        ret[rank] = (losses, model.full_embedding().numpy().copy())
    finally:
        dist.destroy_process_group()


def test_column_sharded_training_step_equals_unsharded(golden):
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    ta = golden['train_array']
    n_layers = 3
    rng = np.random.default_rng(1)
    emb = (rng.standard_normal((nu + ni, 8)) * 0.1).astype(np.float32)
    batch = np.stack([rng.integers(0, nu, 64), rng.integers(0, ni, 64), rng.integers(0, ni, 64)], axis=1).astype(np.int64)
    row, col, val = O.lightgcn_norm_adj(ta, nu, ni)
    a = torch.sparse_coo_tensor(np.stack([row, col]), val, (nu + ni, nu + ni)).to_dense()
    e = torch.nn.Parameter(torch.from_numpy(emb.copy()))
    opt = torch.optim.Adam([e], lr=1e-2)
    b = torch.from_numpy(batch)
    ref_losses = []
    for _ in range(2):
        x, layers = e, [e]
        for _l in range(n_layers):
            x = a @ x
            layers.append(x)
        terms = torch_bpr_terms(torch.stack(layers).mean(0), e, b[:, 0], b[:, 1], b[:, 2], nu)
        loss = terms[0] + 1e-2 * terms[1]
        opt.zero_grad(); loss.backward(); opt.step()
        ref_losses.append(float(loss.detach()))
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_col_worker, args=(world, _free_port(), golden['path'], emb, batch, n_layers, ret), nprocs=world, join=True)
    for r in range(world):
        losses, full = ret[r]
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-5)
        np.testing.assert_allclose(full, e.detach().numpy(), rtol=1e-4, atol=1e-6)


# ---- local blocks built from the pair list (BASELINE config 5's constructor path) ------------------------------
class TorchCsr:
    """What the injected factory hands to cpu_spmm when the blocks arrive as torch tensors (synth.rank_blocks)."""
    def __init__(self, rowptr, col, val, shape, blocks=None):
        self.rowptr, self.col, self.val, self.shape = rowptr.numpy(), col.numpy(), val.numpy(), tuple(shape)
        self.nnz = int(col.shape[0])


def _blocks_worker(rank, world, port, sizes, seed, emb, n_layers, exchange, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from igcn_cf_amd.dist import RowShardedPropagator, ShardLayout
        from igcn_cf_amd.synth import BipartiteGraphDevice
        g = BipartiteGraphDevice(*sizes, 'cpu', seed=seed)          # every rank draws the same graph (same seed) ...
        L = ShardLayout.balanced(g.rowptr_host(), g.n_users, g.n_items, world, fused=exchange == 'fused')
        blocks = g.rank_blocks(L, rank, csr_factory=TorchCsr)       # ... and builds ONLY its own rows of A_hat
        prop = RowShardedPropagator(None, g.n_users, g.n_items, n_layers, rank, world, 'cpu', spmm_fn=cpu_spmm, exchange=exchange,
                                    layout=L, local_blocks=blocks, global_nnz=g.nnz)
        (ulo, uhi), (ilo, ihi) = L.user_rows(rank), L.item_rows(rank)
        nu = g.n_users
        prop.load_local_embedding(torch.from_numpy(emb[ulo:uhi]), torch.from_numpy(emb[nu + ilo:nu + ihi]))
        ru, ri = prop.propagate()
        ret[rank] = (ru[:uhi - ulo].numpy().copy(), ri[:ihi - ilo].numpy().copy(), prop.local_nnz, prop.global_nnz, (ulo, uhi, ilo, ihi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('exchange,n_layers', [('halves', 3), ('halves', 2), ('fused', 3)])
def test_row_sharded_propagation_from_device_built_local_blocks(exchange, n_layers):
    """RowShardedPropagator(layout=, local_blocks=): every rank builds its own blocks from the generator's pair list
    (padded rows, padded column ids computed with torch ops) — no CSR of the whole graph anywhere — and the sharded
    K-layer pass (alternating half order, layer-mean epilogue) equals the unsharded oracle on the same graph."""
    from igcn_cf_amd.synth import BipartiteGraphDevice
    sizes, seed, world = (260, 90, 3000), 5, 2
    g = BipartiteGraphDevice(*sizes, 'cpu', seed=seed)
    nu, ni = g.n_users, g.n_items
    ta = np.stack([g.users.numpy(), g.items.numpy()], axis=1)
    rng = np.random.default_rng(0)
    emb = (rng.standard_normal((nu + ni, 16)) * 0.1).astype(np.float32)
    ref = O.lightgcn_get_rep(O.lightgcn_norm_adj(ta, nu, ni), emb, n_layers)
    ret = mp.Manager().dict()
    mp.spawn(_blocks_worker, args=(world, _free_port(), sizes, seed, emb, n_layers, exchange, ret), nprocs=world, join=True)
    nnz = users_seen = items_seen = 0
    for r in range(world):
        ru, ri, lnnz, gnnz, (ulo, uhi, ilo, ihi) = ret[r]
        np.testing.assert_allclose(ru, ref[ulo:uhi], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ri, ref[nu + ilo:nu + ihi], rtol=1e-5, atol=1e-7)
        nnz += lnnz
        users_seen += uhi - ulo
        items_seen += ihi - ilo
    assert nnz == gnnz == 2 * g.n_edges and users_seen == nu and items_seen == ni


def test_local_blocks_are_checked_against_their_layout():
    from igcn_cf_amd.dist import RowShardedPropagator, ShardLayout
    from igcn_cf_amd.synth import BipartiteGraphDevice
    g = BipartiteGraphDevice(60, 30, 400, 'cpu', seed=1)
    L = ShardLayout.balanced(g.rowptr_host(), g.n_users, g.n_items, 2)
    blocks = g.rank_blocks(L, 0, csr_factory=TorchCsr)
    with pytest.raises(ValueError):                                  # blocks without the layout they were cut with
        RowShardedPropagator(None, 60, 30, 2, 0, 2, 'cpu', spmm_fn=cpu_spmm, exchange='halves', local_blocks=blocks)
    with pytest.raises(ValueError):                                  # a 'halves' layout under the 'fused' exchange
        RowShardedPropagator(None, 60, 30, 2, 0, 2, 'cpu', spmm_fn=cpu_spmm, exchange='fused', layout=L, local_blocks=blocks)
    with pytest.raises(ValueError):                                  # another rank count's blocks
        L3 = ShardLayout.balanced(g.rowptr_host(), g.n_users, g.n_items, 3)
        RowShardedPropagator(None, 60, 30, 2, 0, 2, 'cpu', spmm_fn=cpu_spmm, exchange='halves', layout=L3, local_blocks=blocks)
    with pytest.raises(ValueError):                                  # users and items swapped
        RowShardedPropagator(None, 60, 30, 2, 0, 2, 'cpu', spmm_fn=cpu_spmm, exchange='halves', layout=L, local_blocks=blocks[::-1])


# ---- the exchange logic under RCCL's completion semantics (no node needed) ----------------------------------------------------
from tests.late_collectives import LateCollectives, run_ranks as _run_ranks            # noqa: E402


def _late_pass(golden, world, n_layers, exchange, skip_wait=None):
    from igcn_cf_amd.dist import RowShardedPropagator
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    ta = golden['train_array']
    rng = np.random.default_rng(0)
    emb = (rng.standard_normal((nu + ni, 8)) * 0.1).astype(np.float32)
    coll = LateCollectives(world)
    coll.skip_wait = skip_wait

    def rank_fn(rank):
        prop = RowShardedPropagator(ta, nu, ni, n_layers, rank, world, 'cpu', spmm_fn=cpu_spmm,
                                    csr_factory=lambda rp, c, v, shape: CpuCsr(rp, c, v, shape), exchange=exchange,
                                    collectives=coll.bind(rank))
        L = prop.layout
        (ulo, uhi), (ilo, ihi) = L.user_rows(rank), L.item_rows(rank)
        outs = []
        for _ in range(2):                                      # two passes: the buffers of the first are re-used by the second
            prop.load_local_embedding(torch.from_numpy(emb[ulo:uhi]), torch.from_numpy(emb[nu + ilo:nu + ihi]))
            ru, ri = prop.propagate()
            outs.append((ru[:uhi - ulo].numpy().copy(), ri[:ihi - ilo].numpy().copy()))
        return outs, (ulo, uhi, ilo, ihi)
    res = _run_ranks(world, rank_fn)
    ref = O.lightgcn_get_rep(O.lightgcn_norm_adj(ta, nu, ni), emb, n_layers)
    return res, ref, coll, nu


@pytest.mark.parametrize('world', [2, 3, 8])
@pytest.mark.parametrize('exchange', ['fused', 'halves'])
@pytest.mark.parametrize('n_layers', [1, 2, 3, 4])
def test_exchange_logic_holds_when_collectives_complete_as_late_as_rccl_allows(golden, world, exchange, n_layers):
    """RowShardedPropagator.propagate keeps an all-gather in flight under the other half's SpMM ('halves') and re-uses two
    replicated buffers in turn.  Under gloo a missing or misplaced wait() is invisible (its wait blocks the host and the copy is
    long done); RCCL only orders streams.  Here the copies land at the latest legal moment, destinations are poisoned while a
    collective is in flight and sources are checked at completion: every owned row equals the unsharded oracle, in both of two
    consecutive passes, and nobody touched a source early.  world 2, 3 and 8 (the one size the driver's 8-GPU command uses: on the
    toys a block is a few dozen rows, unequal and padded), K = 1 ... 4, both exchanges."""
    res, ref, coll, nu = _late_pass(golden, world, n_layers, exchange)
    assert not coll.errors, coll.errors
    if exchange == 'halves' and n_layers > 1:
        assert coll.n_async > 0                                  # the overlapped form really was exercised
    for outs, (ulo, uhi, ilo, ihi) in res:
        for ru, ri in outs:
            assert np.isfinite(ru).all() and np.isfinite(ri).all()
            np.testing.assert_allclose(ru, ref[ulo:uhi], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(ri, ref[nu + ilo:nu + ihi], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('nth', [0, 1, 2, 3, 4, 5])
def test_a_forgotten_wait_is_noticed_by_the_late_collectives(golden, nth):
    """Negative control of the harness: the same pass ('halves', K = 3, two ranks) with ONE wait of rank 0 turned into a no-op —
    each of the six asynchronous all-gathers of a pass in turn (the two of the X_0 exchange, the four half-layer gathers).  The
    half-layer that reads the section still in flight sees the poison: rank 0's result is not the oracle's, every time."""
    res, ref, coll, nu = _late_pass(golden, 2, 3, 'halves', skip_wait=(0, nth))
    outs, (ulo, uhi, ilo, ihi) = res[0]
    ru, ri = outs[0]
    assert not (np.isfinite(ru).all() and np.isfinite(ri).all())


@pytest.mark.parametrize('world,exchange', [(2, 'fused'), (3, 'halves'), (8, 'fused'), (8, 'halves')])
def test_sharded_training_steps_under_late_collectives(golden, world, exchange):
    """Two Adam steps of ShardedLightGCN — forward pass, the batch rows' all-reduce (_BatchRowsFn), backward pass through the
    same sharded operator — with every collective completing late: losses and the gathered table equal the dense unsharded
    chain's, world 2, 3 and 8."""
    from igcn_cf_amd.dataset import ProcessedDataset
    from igcn_cf_amd.dist import ShardedLightGCN
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    ta = golden['train_array']
    n_layers = 3
    rng = np.random.default_rng(1)
    emb = (rng.standard_normal((nu + ni, 8)) * 0.1).astype(np.float32)
    batch = np.stack([rng.integers(0, nu, 64), rng.integers(0, ni, 64), rng.integers(0, ni, 64)], axis=1).astype(np.int64)
    row, col, val = O.lightgcn_norm_adj(ta, nu, ni)
    a = torch.sparse_coo_tensor(np.stack([row, col]), val, (nu + ni, nu + ni)).to_dense()
    e = torch.nn.Parameter(torch.from_numpy(emb.copy()))
    opt = torch.optim.Adam([e], lr=1e-2)
    b = torch.from_numpy(batch)
    ref_losses = []
    for _ in range(2):
        x, layers = e, [e]
        for _l in range(n_layers):
            x = a @ x
            layers.append(x)
        terms = torch_bpr_terms(torch.stack(layers).mean(0), e, b[:, 0], b[:, 1], b[:, 2], nu)
        loss = terms[0] + 1e-2 * terms[1]
        opt.zero_grad(); loss.backward(); opt.step()
        ref_losses.append(float(loss.detach()))
    coll = LateCollectives(world)
    ds = ProcessedDataset({'name': 'ProcessedDataset', 'path': golden['path'], 'device': 'cpu'})

    def rank_fn(rank):
        model = ShardedLightGCN(ds, emb.shape[1], n_layers, rank, world, 'cpu', spmm_fn=cpu_spmm,
                                csr_factory=lambda rp, c, v, shape: CpuCsr(rp, c, v, shape), loss_fn=torch_bpr_terms_rows,
                                full_embedding=torch.from_numpy(emb), exchange=exchange, collectives=coll.bind(rank))
        o = torch.optim.Adam(model.parameters(), lr=1e-2)
        losses = []
        for _ in range(2):
            terms = model.bpr_loss_terms(b[:, 0], b[:, 1], b[:, 2])
            l = terms[0] + 1e-2 * terms[1]
            o.zero_grad(); l.backward(); o.step()
            losses.append(float(l))
        return losses, model.full_embedding().numpy().copy()
    res = _run_ranks(world, rank_fn)
    assert not coll.errors, coll.errors
    for losses, full in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-5)
        np.testing.assert_allclose(full, e.detach().numpy(), rtol=1e-4, atol=1e-6)


# ---- world = 8 with blocks that own NOTHING ------------------------------------------------------------------------------------
def _skewed_graph(nu=40, ni=24, seed=5):
    """A tiny bipartite graph in which one item holds most of the edges and three users hold most of the rest: an 8-way
    nnz-balanced cut then has item blocks (and user blocks) with zero rows — ShardLayout.balanced's searchsorted lands several
    boundaries on the same row."""
    rng = np.random.default_rng(seed)
    pairs = {(u, 0) for u in range(nu)}                                  # item 0: every user
    for u in (0, 1, 2):
        pairs |= {(u, i) for i in range(ni)}                             # three users: every item
    for u in range(3, nu):
        pairs.add((u, int(rng.integers(1, ni))))
    return np.array(sorted(pairs), dtype=np.int64), nu, ni


def test_an_eight_way_balanced_cut_may_leave_blocks_empty_and_still_covers_every_row():
    from igcn_cf_amd.dist import ShardLayout
    from igcn_cf_amd.graph import normalized_adjacency_host
    ta, nu, ni = _skewed_graph()
    rowptr, col, val = normalized_adjacency_host(ta, nu, ni)
    for fused in (False, True):
        L = ShardLayout.balanced(rowptr, nu, ni, 8, fused=fused)
        ub, ib = np.diff(L.user_bounds), np.diff(L.item_bounds)
        assert ub.sum() == nu and ib.sum() == ni and (ub >= 0).all() and (ib >= 0).all()
        assert (ib == 0).any(), ib                                       # the case this test is about
        assert L.bu == ub.max() and L.bi == ib.max() and L.n_pad == 8 * (L.bu + L.bi)
        # every node lands in its owner's padded block, no two nodes on the same padded row
        pad = L.pad_index(np.arange(nu + ni))
        assert len(set(pad.tolist())) == nu + ni and pad.max() < L.n_pad
        rank, local = L.owner(np.arange(nu + ni))
        for r in range(8):
            (ulo, uhi), (ilo, ihi) = L.user_rows(r), L.item_rows(r)
            assert (rank[ulo:uhi] == r).all() and (rank[nu + ilo:nu + ihi] == r).all()
            assert (local[ulo:uhi] == np.arange(uhi - ulo)).all() and (local[nu + ilo:nu + ihi] == L.bu + np.arange(ihi - ilo)).all()


@pytest.mark.parametrize('exchange', ['fused', 'halves'])
@pytest.mark.parametrize('n_layers', [1, 3])
def test_eight_ranks_some_owning_nothing_under_late_collectives(exchange, n_layers):
    """The sharded pass over 8 ranks where some ranks own zero item rows (and their padded block is all padding): two passes and
    the gathered table equal the unsharded oracle under the late-completing collectives."""
    from igcn_cf_amd.dist import RowShardedPropagator
    ta, nu, ni = _skewed_graph()
    world = 8
    rng = np.random.default_rng(2)
    emb = (rng.standard_normal((nu + ni, 8)) * 0.1).astype(np.float32)
    coll = LateCollectives(world)

    def rank_fn(rank):
        prop = RowShardedPropagator(ta, nu, ni, n_layers, rank, world, 'cpu', spmm_fn=cpu_spmm,
                                    csr_factory=lambda rp, c, v, shape: CpuCsr(rp, c, v, shape), exchange=exchange,
                                    collectives=coll.bind(rank))
        L = prop.layout
        (ulo, uhi), (ilo, ihi) = L.user_rows(rank), L.item_rows(rank)
        outs = []
        for _ in range(2):
            prop.load_local_embedding(torch.from_numpy(emb[ulo:uhi]), torch.from_numpy(emb[nu + ilo:nu + ihi]))
            ru, ri = prop.propagate()
            outs.append((ru[:uhi - ulo].numpy().copy(), ri[:ihi - ilo].numpy().copy()))
        full = prop.gather_full_rep(ru, ri).numpy().copy()
        return outs, full, (ulo, uhi, ilo, ihi), prop.local_nnz
    res = _run_ranks(world, rank_fn)
    assert not coll.errors, coll.errors
    ref = O.lightgcn_get_rep(O.lightgcn_norm_adj(ta, nu, ni), emb, n_layers)
    empties = 0
    for outs, full, (ulo, uhi, ilo, ihi), _ in res:
        empties += (ihi == ilo) + (uhi == ulo)
        for ru, ri in outs:
            np.testing.assert_allclose(ru, ref[ulo:uhi], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(ri, ref[nu + ilo:nu + ihi], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(full, ref, rtol=1e-5, atol=1e-7)
    assert empties > 0
    assert sum(r[3] for r in res) == 2 * len(ta)

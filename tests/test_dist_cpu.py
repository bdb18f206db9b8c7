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


def _worker(rank, world, port, train_array, nu, ni, emb, n_layers, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from igcn_cf_amd.dist import RowShardedPropagator
        prop = RowShardedPropagator(train_array, nu, ni, n_layers, rank, world, 'cpu', spmm_fn=cpu_spmm,
                                    csr_factory=lambda rp, c, v, shape: CpuCsr(rp, c, v, shape))
        L = prop.layout
        (ulo, uhi), (ilo, ihi) = L.user_rows(rank), L.item_rows(rank)
        prop.load_local_embedding(torch.from_numpy(emb[ulo:uhi]), torch.from_numpy(emb[nu + ilo:nu + ihi]))
        ru, ri = prop.propagate()
        full = prop.gather_full_rep(ru, ri)
        ret[rank] = (ru[:uhi - ulo].numpy().copy(), ri[:ihi - ilo].numpy().copy(), full.numpy().copy(),
                     prop.local_nnz, prop.global_nnz)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_layers', [1, 2, 3, 4])
def test_row_sharded_propagation_equals_unsharded(golden, n_layers):
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    ta = golden['train_array']
    rng = np.random.default_rng(0)
    emb = (rng.standard_normal((nu + ni, 16)) * 0.1).astype(np.float32)
    ref = O.lightgcn_get_rep(O.lightgcn_norm_adj(ta, nu, ni), emb, n_layers)
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ta, nu, ni, emb, n_layers, ret), nprocs=world, join=True)
    from igcn_cf_amd.dist import ShardLayout
    L = ShardLayout(nu, ni, world)
    nnz = 0
    for r in range(world):
        ru, ri, full, lnnz, gnnz = ret[r]
        (ulo, uhi), (ilo, ihi) = L.user_rows(r), L.item_rows(r)
        np.testing.assert_allclose(ru, ref[ulo:uhi], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ri, ref[nu + ilo:nu + ihi], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(full, ref, rtol=1e-5, atol=1e-7)
        nnz += lnnz
    assert nnz == gnnz                                   # every edge owned exactly once


def test_shard_layout_padding():
    from igcn_cf_amd.dist import ShardLayout, local_blocks_host
    L = ShardLayout(5, 3, 4)                              # blocks of 2 users / 1 item, ranks with empty tails
    assert (L.bu, L.bi, L.pu, L.pi) == (2, 1, 8, 4)
    assert L.user_rows(2) == (4, 5) and L.user_rows(3) == (6, 5) and L.item_rows(3) == (3, 3)
    np.testing.assert_array_equal(L.pad_index([0, 4, 5, 7]), [0, 4, 8, 10])
    rowptr = np.arange(9, dtype=np.int64)                 # one entry per row
    col = np.array([5, 6, 7, 5, 6, 0, 1, 2], dtype=np.int32)
    val = np.ones(8, dtype=np.float32)
    (urp, ucol, _), (irp, icol, _) = local_blocks_host(rowptr, col, val, L, 3)
    assert urp.tolist() == [0, 0, 0] and irp.tolist() == [0, 0]          # rank 3 owns padding only
    (urp, ucol, _), (irp, icol, _) = local_blocks_host(rowptr, col, val, L, 2)
    assert urp.tolist() == [0, 1, 1] and ucol.tolist() == [9] and irp.tolist() == [0, 1] and icol.tolist() == [2]


# ---- sharded training step (forward + backward + Adam) vs the same step unsharded ----------------
def torch_bpr_terms(rep, emb, users, pos, neg, n_users):
    """trainer.py:238-243 + model.py:110-116 in torch (checker implementation for the CPU test)."""
    ur, pr, nr = rep[users], rep[n_users + pos], rep[n_users + neg]
    ue, pe, ne = emb[users], emb[n_users + pos], emb[n_users + neg]
    bpr = torch.nn.functional.softplus((ur * nr).sum(1) - (ur * pr).sum(1)).mean()
    l2 = ((ue ** 2).sum(1) + (pe ** 2).sum(1) + (ne ** 2).sum(1)).mean()
    return torch.stack([bpr, l2])


def _train_worker(rank, world, port, path, emb, batch, n_layers, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from igcn_cf_amd.dataset import ProcessedDataset
        from igcn_cf_amd.dist import ShardedLightGCN
        ds = ProcessedDataset({'name': 'ProcessedDataset', 'path': path, 'device': 'cpu'})
        model = ShardedLightGCN(ds, emb.shape[1], n_layers, rank, world, 'cpu', spmm_fn=cpu_spmm,
                                csr_factory=lambda rp, c, v, shape: CpuCsr(rp, c, v, shape), loss_fn=torch_bpr_terms,
                                full_embedding=torch.from_numpy(emb))
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


def test_sharded_training_step_equals_unsharded(golden):
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
    mp.spawn(_train_worker, args=(world, _free_port(), golden['path'], emb, batch, n_layers, ret), nprocs=world, join=True)
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

"""The row-sharded PRODUCT path (HIP SpMM, fused HIP BPR loss, HIP top-k; igcn_cf_amd/dist.py) with more than one
rank on the GPU box: the ranks are separate processes that share cuda:0 and exchange over gloo (device tensors,
staged through the host by gloo).  RCCL itself refuses two ranks on one device; its world-size-1 path is covered in
test_models_gpu.py and the partitioning logic under gloo on the CPU in test_dist_cpu.py.  What this adds: the HIP
kernels on rank-local CSR blocks with padded global column ids, unequal (nnz-balanced) blocks, both exchanges,
forward + backward + Adam and the user-sharded evaluation, against the unsharded float64 chain."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, path, emb, batch, n_layers, exchange, k, excl, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from igcn_cf_amd.dataset import ProcessedDataset
        from igcn_cf_amd.dist import ShardedLightGCN
        dev = torch.device('cuda', 0)
        ds = ProcessedDataset({'name': 'ProcessedDataset', 'path': path, 'device': dev})
        model = ShardedLightGCN(ds, emb.shape[1], n_layers, rank, world, dev, full_embedding=torch.from_numpy(emb),
                                exchange=exchange)
        L = model.prop.layout
        (ulo, uhi), (ilo, ihi) = L.user_rows(rank), L.item_rows(rank)
        with torch.no_grad():
            ru, ri = model.get_rep_local()
            rep0 = (ru[:uhi - ulo].cpu().numpy().copy(), ri[:ihi - ilo].cpu().numpy().copy())
            rec = model.recommend_local(k, excl=excl).cpu().numpy().copy()
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        b = torch.from_numpy(batch).to(dev)
        losses = []
        for _ in range(2):
            terms = model.bpr_loss_terms(b[:, 0].contiguous(), b[:, 1].contiguous(), b[:, 2].contiguous())
            loss = terms[0] + 1e-2 * terms[1]
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(float(loss.detach()))
        ret[rank] = (rep0, rec, losses, model.full_embedding().cpu().numpy().copy(), (ulo, uhi, ilo, ihi), model.prop.exchange)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,exchange,d', [(2, 'halves', 16), (3, 'fused', 16), (2, 'halves', 128)])
def test_row_sharded_product_path_ranks_share_one_gpu(golden, world, exchange, d):
    nu, ni = int(golden['n_users']), int(golden['n_items'])
    ta = golden['train_array']
    n_layers, k = 3, 5          # d = 128: the one-row-per-wave SpMM variant and the exchange BASELINE config 5 uses
    rng = np.random.default_rng(3)
    emb = (rng.standard_normal((nu + ni, d)) * 0.1).astype(np.float32)
    batch = np.stack([rng.integers(0, nu, 96), rng.integers(0, ni, 96), rng.integers(0, ni, 96)], axis=1).astype(np.int64)
    batch[5] = batch[4]                                                      # duplicate ids in the batch
    # masked items per user = the train lists (trainer.py:153-158), as a host CSR over all users
    order = np.lexsort((ta[:, 1], ta[:, 0]))
    excl_col = ta[order, 1].astype(np.int32)
    excl_rowptr = np.concatenate([[0], np.cumsum(np.bincount(ta[:, 0], minlength=nu))]).astype(np.int64)

    # unsharded float64 chain: representation, two Adam steps, masked top-k
    row, col, val = O.lightgcn_norm_adj(ta, nu, ni)
    a = torch.sparse_coo_tensor(np.stack([row, col]), val.astype(np.float64), (nu + ni, nu + ni)).to_dense()
    e = torch.nn.Parameter(torch.from_numpy(emb.astype(np.float64)))

    def rep_of(e):
        x, layers = e, [e]
        for _ in range(n_layers):
            x = a @ x
            layers.append(x)
        return torch.stack(layers).mean(0)
    rep_ref = rep_of(e).detach().numpy()
    scores = rep_ref[:nu] @ rep_ref[nu:].T
    scores[ta[:, 0], ta[:, 1]] = -np.inf
    opt = torch.optim.Adam([e], lr=1e-2)
    b = torch.from_numpy(batch)
    ref_losses = []
    for _ in range(2):
        rep = rep_of(e)
        ur, pr, nr = rep[b[:, 0]], rep[nu + b[:, 1]], rep[nu + b[:, 2]]
        l2 = ((e[b[:, 0]] ** 2).sum(1) + (e[nu + b[:, 1]] ** 2).sum(1) + (e[nu + b[:, 2]] ** 2).sum(1)).mean()
        loss = torch.nn.functional.softplus((ur * nr).sum(1) - (ur * pr).sum(1)).mean() + 1e-2 * l2
        opt.zero_grad(); loss.backward(); opt.step()
        ref_losses.append(float(loss.detach()))

    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), golden['path'], emb, batch, n_layers, exchange, k,
                            (excl_rowptr, excl_col), ret), nprocs=world, join=True)
    users_seen = 0
    for r in range(world):
        (ru, ri), rec, losses, full, (ulo, uhi, ilo, ihi), used = ret[r]
        assert used == exchange
        np.testing.assert_allclose(ru, rep_ref[ulo:uhi], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ri, rep_ref[nu + ilo:nu + ihi], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-5)
        # two Adam steps: an update is lr * m / sqrt(v), so rounding-level gradient differences show at the 1e-5 level
        np.testing.assert_allclose(full, e.detach().numpy(), rtol=1e-4, atol=2e-5)
        assert rec.shape == (uhi - ulo, k)
        for j, u in enumerate(range(ulo, uhi)):                              # same score multiset as the float64 ranking
            want = np.sort(scores[u])[::-1][:k]
            np.testing.assert_allclose(scores[u, rec[j]], want, rtol=1e-4, atol=1e-6)
        users_seen += uhi - ulo
    assert users_seen == nu

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


# ------------------------------------------------------------------------------------------------------------------
# BASELINE config 4 at its real size: LightGCN 3-layer d = 64 on the Amazon-book-like split, rows sharded over 2 and 4
# ranks (here sharing the one GPU, exchange over gloo), against the UNSHARDED HIP path on the same weights.
# ------------------------------------------------------------------------------------------------------------------
def _amazon_worker(rank, world, port, seed, batch, k, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from igcn_cf_amd.dataset import SyntheticDataset
        from igcn_cf_amd.dist import ShardedLightGCN
        from igcn_cf_amd.trainer import _merge_sorted_csr
        dev = torch.device('cuda', 0)
        ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021, 'device': dev})
        model = ShardedLightGCN(ds, 64, 3, rank, world, dev, seed=seed)           # the same N(0, 0.1^2) table on every rank
        L = model.prop.layout
        (ulo, uhi), (ilo, ihi) = L.user_rows(rank), L.item_rows(rank)
        excl = _merge_sorted_csr(ds.csr('train', sort=True), ds.csr('val', sort=True))
        with torch.no_grad():
            ru, ri = model.get_rep_local()
            rep_u, rep_i = ru[:uhi - ulo:53].cpu().numpy().copy(), ri[:ihi - ilo:53].cpu().numpy().copy()
            rec = model.recommend_local(k, excl=excl)[::29].cpu().numpy().copy()
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        b = torch.from_numpy(batch).to(dev)
        losses = []
        for _ in range(2):
            terms = model.bpr_loss_terms(b[:, 0].contiguous(), b[:, 1].contiguous(), b[:, 2].contiguous())
            loss = terms[0] + 1e-2 * terms[1]
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(float(loss.detach()))
        emb_u = model.emb_users.detach()[:uhi - ulo:53].cpu().numpy().copy()
        emb_i = model.emb_items.detach()[:ihi - ilo:53].cpu().numpy().copy()
        ret[rank] = dict(bounds=(ulo, uhi, ilo, ihi), rep_u=rep_u, rep_i=rep_i, rec=rec, losses=losses, emb_u=emb_u, emb_i=emb_i,
                         exchange=model.prop.exchange, local_nnz=model.prop.local_nnz)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_config4_amazon_size_row_sharded_against_the_unsharded_hip_path(world):
    from igcn_cf_amd import ops
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.model import get_model
    from igcn_cf_amd.trainer import _merge_sorted_csr, _csr_to_device
    k, seed = 20, 77
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021, 'device': 'cuda'})
    nu, ni = ds.n_users, ds.n_items
    rng = np.random.default_rng(3)
    B = 2048
    batch = np.stack([rng.integers(0, nu, B), rng.integers(0, ni, B), rng.integers(0, ni, B)], axis=1).astype(np.int64)
    batch[5] = batch[4]
    # the unsharded product path on the same table
    model = get_model({'name': 'LightGCN', 'embedding_size': 64, 'n_layers': 3, 'device': 'cuda'}, ds)
    g = torch.Generator(device='cpu').manual_seed(seed)
    with torch.no_grad():
        model.embedding.weight.copy_((torch.randn(nu + ni, 64, generator=g) * 0.1).cuda())
    model.eval()
    with torch.no_grad():
        rep = model.get_rep().clone()
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    b = torch.from_numpy(batch).cuda()
    ref_losses = []
    for _ in range(2):
        terms = model.bpr_loss_terms(b[:, 0].contiguous(), b[:, 1].contiguous(), b[:, 2].contiguous())
        loss = terms[0] + 1e-2 * terms[1]
        opt.zero_grad(); loss.backward(); opt.step()
        ref_losses.append(float(loss.detach()))
    emb2 = model.embedding.weight.detach()
    excl = _merge_sorted_csr(ds.csr('train', sort=True), ds.csr('val', sort=True))
    erp, ecl = _csr_to_device(excl[0], excl[1], 'cuda')

    ret = mp.Manager().dict()
    mp.spawn(_amazon_worker, args=(world, _free_port(), seed, batch, k, ret), nprocs=world, join=True)
    users_seen = nnz_seen = 0
    for r in range(world):
        out = ret[r]
        ulo, uhi, ilo, ihi = out['bounds']
        assert out['exchange'] == 'fused'                                         # 52.8 MB operand: one all-gather per layer
        scale = float(rep.abs().max())
        assert np.abs(out['rep_u'] - rep[ulo:uhi:53].cpu().numpy()).max() <= 1e-5 * scale
        assert np.abs(out['rep_i'] - rep[nu + ilo:nu + ihi:53].cpu().numpy()).max() <= 1e-5 * scale
        np.testing.assert_allclose(out['losses'], ref_losses, rtol=2e-6)
        # two Adam steps (lr 1e-2): an update is lr * m / sqrt(v); rounding-level gradient differences show at 1e-5
        np.testing.assert_allclose(out['emb_u'], emb2[ulo:uhi:53].cpu().numpy(), rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(out['emb_i'], emb2[nu + ilo:nu + ihi:53].cpu().numpy(), rtol=1e-4, atol=3e-5)
        # user-sharded top-20 (train + val lists masked): the same score multiset as the unsharded fused scorer's
        users = torch.arange(ulo, uhi, 29, device='cuda')
        idx, val = ops.score_topk(rep, rep[nu:], k, user_ids=users, excl_rowptr=erp, excl_col=ecl, mode='exact')
        got = (rep[users][:, None, :] * rep[nu:][torch.from_numpy(out['rec']).cuda()]).sum(-1)
        assert float((got - val).abs().max()) <= 1e-5 * float(val.abs().max())
        assert float((torch.from_numpy(out['rec']).cuda() == idx).float().mean()) > 0.999      # ids: equal up to near-ties
        users_seen += uhi - ulo
        nnz_seen += out['local_nnz']
    assert users_seen == nu and nnz_seen == model.norm_adj.nnz
    nnzs = [ret[r]['local_nnz'] for r in range(world)]
    assert max(nnzs) / (sum(nnzs) / world) < 1.02                                 # nnz-balanced blocks

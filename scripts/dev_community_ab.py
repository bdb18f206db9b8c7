"""Round 4: does node relabelling pay once the graph HAS community structure?  One SpMM launch (d = 64, XCD
plan, the production kernel) on
  (1) the SURVEY 8(d) Amazon-like graph (item popularity is its only structure) and
  (2) the same sizes / degree laws with PLANTED communities (dataset.SyntheticDataset communities = 64, community_share 0.8),
each in the original labels, relabelled by a cheap bipartite co-clustering of the graph (label propagation from the top-degree
items — the labels carry nothing, ids are random), and — (2) only — relabelled by the planted communities (what a perfect
clustering would reach).  The relabelled product runs behind a permutation: y = P^T (A' (P x)), compared with the original
labels' result (<= 1e-5: a row's nonzeros are summed in another order).  With PMC=1 (under rocprofv3 --pmc) every variant is
launched N times in a fixed order for scripts/dev_pmc_by_variant.py; otherwise same-process timings, five rounds.
Prints JSON lines."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm
from scripts.dev_spmm_bench import time_ms

PMC = os.environ.get('PMC') == '1'
N_LAUNCH = 4
D = 64


def co_cluster(ta, nu, ni, n_seeds=512, rounds=6):
    """Label propagation on the bipartite graph: the n_seeds items of highest degree start with a label each; users take
    the label most of their items carry, items the one most of their users carry; a few rounds.  Host, scipy: ~1 s per
    round at Amazon size — a one-off per graph, like the XCD plan."""
    import scipy.sparse as sp
    r = sp.csr_matrix((np.ones(len(ta), dtype=np.float32), (ta[:, 0], ta[:, 1])), shape=(nu, ni))
    rt = r.T.tocsr()
    deg_i = np.asarray(r.sum(0)).ravel()
    seeds = np.argsort(-deg_i, kind='stable')[:n_seeds]
    item_lab = np.full(ni, -1, dtype=np.int64)
    item_lab[seeds] = np.arange(n_seeds)
    user_lab = np.full(nu, -1, dtype=np.int64)

    def vote(mat, lab, n_out):
        known = lab >= 0
        onehot = sp.csr_matrix((np.ones(known.sum(), dtype=np.float32), (np.flatnonzero(known), lab[known])), shape=(lab.shape[0], n_seeds))
        votes = (mat @ onehot).toarray()
        out = votes.argmax(1)
        out[votes.max(1) == 0] = -1
        return out
    for _ in range(rounds):
        user_lab = vote(r, item_lab, nu)
        new_items = vote(rt, user_lab, ni)
        new_items[seeds] = np.arange(n_seeds)               # seeds keep their labels
        item_lab = new_items
    return user_lab, item_lab


def order_by(lab, deg):
    """perm[new] = old: nodes of one label adjacent (unlabelled ones last), by descending degree inside a label."""
    key = np.where(lab < 0, lab.max() + 1, lab)
    return np.lexsort((-deg, key))


def relabel(ta, nu, ni, perm_u, perm_i):
    inv_u = np.empty(nu, dtype=np.int64); inv_u[perm_u] = np.arange(nu)
    inv_i = np.empty(ni, dtype=np.int64); inv_i[perm_i] = np.arange(ni)
    return np.stack([inv_u[ta[:, 0]], inv_i[ta[:, 1]]], axis=1)


def build(ta, nu, ni):
    rowptr, col, val = normalized_adjacency_host(ta, nu, ni)
    return CsrMatrix(rowptr, col, val, (nu + ni, nu + ni), 'cuda', order_blocks=[0, nu, nu + ni], xcd_plan=XCD_PLAN)


def main():
    graphs = {'survey_8d': {}, 'planted_64_communities': {'communities': 64, 'community_share': 0.8}}
    gen = torch.Generator(device='cuda').manual_seed(3)
    plan = []                                                # (name, csr, x in the variant's labels, check)
    for gname, extra in graphs.items():
        ds = SyntheticDataset(dict({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021, 'device': 'cuda'}, **extra))
        nu, ni, ta = ds.n_users, ds.n_items, ds.train_array
        n = nu + ni
        deg_u, deg_i = np.bincount(ta[:, 0], minlength=nu), np.bincount(ta[:, 1], minlength=ni)
        x = torch.randn(n, D, device='cuda', generator=gen) * 0.1
        base = build(ta, nu, ni)
        y0 = spmm(base, x)
        plan.append((gname + '/original_labels', base, x, None))
        t0 = time.perf_counter()
        ul, il = co_cluster(ta, nu, ni)
        t_cluster = time.perf_counter() - t0
        perms = {'co_clustered': (order_by(ul, deg_u), order_by(il, deg_i))}
        info = {'graph': gname, 'nnz': base.nnz, 'co_clustering_s': round(t_cluster, 2), 'labels_found_users': int((ul >= 0).mean() * 1000) / 1000,
                'distinct_user_labels': int(len(np.unique(ul[ul >= 0])))}
        if extra:
            perms['planted_communities'] = (order_by(ds.user_community, deg_u), order_by(ds.item_community, deg_i))
            # how well the clustering recovers what was planted: share of users whose label's majority community is their own
            ok = ul >= 0
            tab = np.zeros((ul.max() + 1, 64), dtype=np.int64)
            np.add.at(tab, (ul[ok], ds.user_community[ok]), 1)
            info['users_in_their_labels_majority_community'] = round(float(tab.max(1).sum() / ok.sum()), 3)
            info['in_community_edge_share'] = round(float((ds.user_community[ta[:, 0]] == ds.item_community[ta[:, 1]]).mean()), 3)
        for pname, (pu, pi) in perms.items():
            m = build(relabel(ta, nu, ni, pu, pi), nu, ni)
            p = torch.from_numpy(np.concatenate([pu, nu + pi])).cuda()          # new row -> old row
            xp = x[p].contiguous()
            yp = spmm(m, xp)
            back = torch.empty_like(yp); back[p] = yp                           # un-permute on output
            err = float((back - y0).abs().max() / y0.abs().max())
            info['rel_err_' + pname] = err
            assert err <= 1e-5, (gname, pname, err)
            plan.append((gname + '/' + pname, m, xp, None))
        print(json.dumps(info), flush=True)
    y = torch.empty(plan[0][1].shape[0], D, device='cuda')
    if PMC:
        print(json.dumps({'pmc_order': [p[0] for p in plan], 'launches_per_variant': N_LAUNCH}), flush=True)
        for name, m, xv, _ in plan:
            for _ in range(N_LAUNCH):
                spmm(m, xv, out=y)
        torch.cuda.synchronize()
        return
    res = {p[0]: [] for p in plan}
    for _ in range(5):
        for name, m, xv, _ in plan:
            res[name].append(time_ms(lambda: spmm(m, xv, out=y), reps=50))
    print(json.dumps({'ms_per_launch_median_of_5': {k: round(sorted(v)[2], 4) for k, v in res.items()}}), flush=True)


if __name__ == '__main__':
    main()

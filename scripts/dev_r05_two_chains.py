"""Developer experiment: the K = 3 pass as two independent chains on two streams.

A_hat is bipartite: user rows gather item rows only and vice versa, so with the factored mean  T = A X0,  U = X0 + A T,
out = (U + A U) / 4  the half-launches form two chains that meet only before the last step:
    chain A:  T_i = A_i X0   ->  U_u = X0_u + A_u T      (A_u gathers item rows = T_i only)
    chain B:  T_u = A_u X0   ->  U_i = X0_i + A_i T      (A_i gathers user rows = T_u only)
    join, then  out_u = (U_u + A_u U) / 4  on one stream,  out_i = (U_i + A_i U) / 4  on the other.
No launch of one chain waits for the other chain's long-row reduce kernel or tail until the join.  Timed against the one-stream
pass (ops.propagate_mean), interleaved; results compared.  (Eager launches: the two-stream form pays Python stream switches on top;
capturing it as a multi-stream HIP graph crashed the host process on ROCm 7.2 and was not pursued — the one-stream halves form,
which has no such overhead, is already 5 % slower than one launch per layer.)"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
from igcn_cf_amd.ops import spmm, propagate_mean
from scripts.dev_spmm_bench import time_ms

for preset in ('amazon', 'gowalla', 'yelp'):
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021})
    nu, n = ds.n_users, ds.n_users + ds.n_items
    ni = n - nu
    rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ds.n_items)
    whole = CsrMatrix(rowptr, col, val, (n, n), 'cuda', order_blocks=[0, nu, n], xcd_plan=XCD_PLAN)
    eu = int(rowptr[nu])

    def half(lo, hi):
        rp = (rowptr[lo:hi + 1] - rowptr[lo]).copy()
        return CsrMatrix(rp, col[rowptr[lo]:rowptr[hi]].copy(), val[rowptr[lo]:rowptr[hi]].copy(), (hi - lo, n), 'cuda',
                         order_blocks=[0, hi - lo], xcd_plan=XCD_PLAN)
    Au = [half(0, nu) for _ in range(2)]          # (two copies each: a matrix's partial-sum workspace belongs to one stream at a time)
    Ai = [half(nu, n) for _ in range(2)]
    x0 = torch.randn(n, 64, device='cuda') * 0.1
    T, U, out = (torch.empty_like(x0) for _ in range(3))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    e1, e2, ea, eb = (torch.cuda.Event() for _ in range(4))

    def two_chains():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            spmm(Ai[0], x0, out=T[nu:])
            spmm(Au[0], T, out=U[:nu], adds=[x0[:nu]])
            e1.record(s1)
        with torch.cuda.stream(s2):
            spmm(Au[1], x0, out=T[:nu])
            spmm(Ai[1], T, out=U[nu:], adds=[x0[nu:]])
            e2.record(s2)
        with torch.cuda.stream(s1):
            s1.wait_event(e2)
            spmm(Au[0], U, out=out[:nu], adds=[U[:nu]], out_scale=0.25, add_scale=0.25)
        with torch.cuda.stream(s2):
            s2.wait_event(e1)
            spmm(Ai[1], U, out=out[nu:], adds=[U[nu:]], out_scale=0.25, add_scale=0.25)
        cur.wait_stream(s1); cur.wait_stream(s2)
        return out

    def halves_one_stream():
        spmm(Ai[0], x0, out=T[nu:]); spmm(Au[0], x0, out=T[:nu])
        spmm(Au[0], T, out=U[:nu], adds=[x0[:nu]]); spmm(Ai[0], T, out=U[nu:], adds=[x0[nu:]])
        spmm(Au[0], U, out=out[:nu], adds=[U[:nu]], out_scale=0.25, add_scale=0.25)
        spmm(Ai[0], U, out=out[nu:], adds=[U[nu:]], out_scale=0.25, add_scale=0.25)
        return out
    ref = propagate_mean(whole, x0, 3).clone()

    got = two_chains().clone()
    torch.cuda.synchronize()
    res = {'one_launch_per_layer': [], 'two_chains': [], 'halves_one_stream': []}
    for rnd in range(5):
        res['one_launch_per_layer'].append(time_ms(lambda: propagate_mean(whole, x0, 3), 100, 20))
        res['two_chains'].append(time_ms(two_chains, 100, 20))
        res['halves_one_stream'].append(time_ms(halves_one_stream, 100, 20))
    print(json.dumps({'preset': preset, 'ms': {k: round(sorted(v)[2], 4) for k, v in res.items()},
                      'max_abs_diff': float((got - ref).abs().max()), 'equal': bool(torch.equal(got, ref))}), flush=True)

"""BASELINE config 2 (LightGCN 3-layer d = 64 on the Gowalla-like split): where do the 0.11 ms of a get_rep pass go, and what
cuts them?  (the layer loop replaced is model.py:96-106.)

A pass = 3 SpMM launches + 3 long-row reduce launches.  Same process, interleaved rounds, HIP events:
  * the default plan (XCD plan, rows above 112 nonzeros cut) against plans that cut fewer rows (thresholds 256 ... 4096) and the
    plain long-row plan — fewer segments, and from some threshold on NO cut row and no reduce launch at all;
  * each of them eager, with the cut rows added up inside the launch (per-call knob fold = 1: no reduce launch), and as ONE
    captured HIP graph replayed (the host then issues one call per pass instead of six);
  * kernel-only time of the pass (sum of its kernels alone would need a profiler; here: the pass with the host far ahead —
    20 passes enqueued per timing — so that host launch cost cannot be what is measured).
Prints one JSON line per variant; also Yelp-like and Amazon-like for the graph-capture question."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import ops
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host

dev = torch.device('cuda', 0)
d, K = 64, 3
presets = sys.argv[1:] or ['gowalla', 'yelp', 'amazon']


def timed(fn, reps=400, warm=50):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3                # us


for preset in presets:
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021, 'device': dev})
    n = ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
    lens = np.diff(rowptr)
    g = torch.Generator(device='cpu').manual_seed(2021)
    x = (torch.randn(n, d, generator=g) * 0.1).to(dev)
    blocks = [0, ds.n_users, n]
    plans = [('xcd112_default', dict(xcd_plan={'threshold': 112}))]
    if preset == 'gowalla':
        plans += [('xcd%d' % t, dict(xcd_plan={'threshold': t, 'segment_len': 256})) for t in (256, 512, 1024, 2048, 1 << 20)]
        plans += [('plain256', dict()), ('plain_uncut', dict(long_threshold=1 << 20, segment_len=1 << 20))]
    variants = {}
    ref = None
    for pname, kw in plans:
        csr = CsrMatrix(rowptr, col, val, (n, n), dev, order_blocks=blocks, **kw)
        y = ops.propagate_mean(csr, x, K)
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max() / ref.abs().max())
        out = torch.empty_like(x)
        fns = {'eager': lambda csr=csr: ops.propagate_mean(csr, x, K, out=out)}
        if csr.n_long and getattr(csr, 'closing_segments', False):
            fns['eager_fold'] = lambda csr=csr: ops.propagate_mean(csr, x, K, out=out, tune={'fold': 1})
        graphs = {}
        for key in list(fns):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fns[key]()
            torch.cuda.current_stream().wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                fns[key]()
            graphs[key.replace('eager', 'hip_graph')] = gr
        for key, gr in graphs.items():
            fns[key] = gr.replay
        variants[pname] = (csr, fns, err)
    # interleaved rounds: every variant once per round, three rounds, the median
    samples = {(p, k): [] for p, (_, fns, _) in variants.items() for k in fns}
    for _ in range(3):
        for p, (_, fns, _) in variants.items():
            for k, fn in fns.items():
                samples[(p, k)].append(timed(fn))
    for p, (csr, fns, err) in variants.items():
        rec = {'preset': preset, 'plan': p, 'rows': n, 'nnz': int(rowptr[-1]), 'max_row': int(lens.max()), 'cut_rows': csr.n_long,
               'segments': csr.n_segments, 'rel_err_vs_default_plan': err}
        for k in fns:
            rec[k + '_us'] = round(sorted(samples[(p, k)])[1], 2)
        print(json.dumps(rec), flush=True)
    del variants
    torch.cuda.empty_cache()

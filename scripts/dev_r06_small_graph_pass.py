"""BASELINE config 2 (LightGCN 3-layer d = 64 on the Gowalla-like split): where do the 0.11 ms of a get_rep pass go, and what
cuts them?  (the layer loop replaced is model.py:96-106.)

A pass = 3 SpMM launches + 3 long-row reduce launches.  Same process, interleaved rounds, HIP events:
  * the default plan (XCD plan, rows above 112 nonzeros cut) against plans that cut fewer rows (thresholds 256 ... 4096) and the
    plain long-row plan — fewer segments, and from some threshold on NO cut row and no reduce launch at all;
  * each of them eager, with the cut rows added up inside the launch (per-call knob fold = 1: no reduce launch), and as ONE
    captured HIP graph replayed (the host then issues one call per pass instead of six);
  * kernel-only time of the pass (sum of its kernels alone would need a profiler; here: the pass with the host far ahead —
    20 passes enqueued per timing — so that host launch cost cannot be what is measured).
Prints one JSON line per variant; also Yelp-like and Amazon-like for the graph-capture question.
--finer: the opposite direction instead — MORE cuts (thresholds 48 ... 160) and other grids (per-call knob blocks_per_cu), eager only.
(The first version of this script bound `out` late in its lambdas: the captured graphs of earlier plans then wrote into an `out`
tensor that had been freed — a GPU memory fault in the script, not in the library; every lambda now holds its own tensors.)"""
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
presets = [a for a in sys.argv[1:] if not a.startswith('--')] or ['gowalla', 'yelp', 'amazon']
TRACE = '--trace' in sys.argv          # synchronize + a progress line after every step (locating a fault)


def step(msg):
    if TRACE:
        torch.cuda.synchronize()
        sys.stderr.write('step ok: %s\n' % msg)
        sys.stderr.flush()


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
    if preset == 'gowalla' and '--finer' not in sys.argv:
        plans += [('xcd%d' % t, dict(xcd_plan={'threshold': t, 'segment_len': 256})) for t in (256, 512, 1024, 2048, 1 << 20)]
        plans += [('plain256', dict()), ('plain_uncut', dict(long_threshold=1 << 20, segment_len=1 << 20))]
    if '--finer' in sys.argv:            # second question: do MORE cuts / another grid help a small graph?
        plans += [('xcd%d' % t, dict(xcd_plan={'threshold': t})) for t in (48, 64, 80, 96, 128, 160)]
    variants = {}
    ref = None
    for pname, kw in plans:
        csr = CsrMatrix(rowptr, col, val, (n, n), dev, order_blocks=blocks, **kw)
        step('%s %s: plan built (cut rows %d, segments %d)' % (preset, pname, csr.n_long, csr.n_segments))
        y = ops.propagate_mean(csr, x, K)
        step('%s %s: eager pass' % (preset, pname))
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max() / ref.abs().max())
        out = torch.empty_like(x)
        fns = {'eager': lambda csr=csr, out=out: ops.propagate_mean(csr, x, K, out=out)}
        if csr.n_long and getattr(csr, 'closing_segments', False):
            fns['eager_fold'] = lambda csr=csr, out=out: ops.propagate_mean(csr, x, K, out=out, tune={'fold': 1})
        if '--finer' in sys.argv and pname == 'xcd112_default':
            for b in (6, 8, 12, 16, 24, 32, 48):
                fns['eager_grid_%d_per_cu' % b] = lambda csr=csr, out=out, b=b: ops.propagate_mean(csr, x, K, out=out, tune={'blocks_per_cu': b})
        for key in fns:
            fns[key]()
            step('%s %s: %s with out=' % (preset, pname, key))
            assert float((out - ref).abs().max() / ref.abs().max()) < 1e-5, (pname, key)
        graphs = {}
        for key in [k for k in fns if '--finer' not in sys.argv and '_grid_' not in k]:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fns[key]()
            torch.cuda.current_stream().wait_stream(side)
            step('%s %s: %s warm-up on the side stream' % (preset, pname, key))
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                fns[key]()
            step('%s %s: %s captured' % (preset, pname, key))
            gr.replay()
            step('%s %s: %s replayed once' % (preset, pname, key))
            graphs[key.replace('eager', 'hip_graph')] = gr
        for key, gr in graphs.items():
            fns[key] = gr.replay
        variants[pname] = (csr, fns, err)
    # interleaved rounds: every variant once per round, three rounds, the median
    samples = {(p, k): [] for p, (_, fns, _) in variants.items() for k in fns}
    for rnd in range(3):
        for p, (_, fns, _) in variants.items():
            for k, fn in fns.items():
                samples[(p, k)].append(timed(fn))
                step('%s %s: %s timed (round %d)' % (preset, p, k, rnd))
    for p, (csr, fns, err) in variants.items():
        rec = {'preset': preset, 'plan': p, 'rows': n, 'nnz': int(rowptr[-1]), 'max_row': int(lens.max()), 'cut_rows': csr.n_long,
               'segments': csr.n_segments, 'rel_err_vs_default_plan': err}
        for k in fns:
            rec[k + '_us'] = round(sorted(samples[(p, k)])[1], 2)
        print(json.dumps(rec), flush=True)
    del variants
    torch.cuda.empty_cache()

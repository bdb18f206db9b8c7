"""What do the gathers of the HOTTEST item rows cost the user-row phase of the SpMM today, and where are they served from?
(the costing behind north_star's "LDS-staged embedding tiles"; the line replaced is model.py:102.)

An LDS-resident copy of the top-H item rows could at best make those gathers free.  This program measures exactly that upper
bound on the production kernel: the user-row phase (row_mask = user rows) of one A_hat launch on the seeded Amazon-like and
Yelp-like graphs, d = 64, XCD plan — once with every gather, and once per H in (80, 320, 640) with the top-H items' columns
masked out (col_mask: a masked edge issues no gather and no FMA — the launch that an ideal, free LDS copy would leave).  Under
rocprofv3 (scripts/dev_r06_hot_rows_l2.sh) the TCC_HIT / TCC_MISS deltas between the full launch and a masked one are the L2
requests of the hot gathers alone: their L2 hit rate, and — against 2 lines per gathered 256-byte row — how many never reached
L2 at all (served by the CU's vector L1).  Each variant N_LAUNCH times in this order, nothing else of the main kernel between.
    python scripts/dev_r06_hot_rows_l2.py                 (on the GPU box; prints one JSON line with sizes and HIP-event times)
    python scripts/dev_r06_hot_rows_l2.py summarize TAG   (here: gpurun_out/hot_TAG/ -> profiles/r06c_hot_item_rows_l2.json)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_LAUNCH = 6
HS = (80, 320, 640)
KERNEL = 'spmm_csr_multirow_kernel<16'


def run():
    import numpy as np
    import torch
    from igcn_cf_amd import ops
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.graph import XCD_PLAN, CsrMatrix, normalized_adjacency_host
    dev = torch.device('cuda', 0)
    d = 64
    out = {'d': d, 'n_launch': N_LAUNCH, 'order': [], 'graphs': {}, 'launches': {}}
    variants = []
    keep = []
    for preset in ('amazon', 'yelp'):
        ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset, 'seed': 2021, 'device': dev})
        n = ds.n_users + ds.n_items
        rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
        csr = CsrMatrix(rowptr, col, val, (n, n), dev, order_blocks=[0, ds.n_users, n], xcd_plan=XCD_PLAN)
        g = torch.Generator(device='cpu').manual_seed(2021)
        x = (torch.randn(n, d, generator=g) * 0.1).to(dev)
        y = torch.empty_like(x)
        ta = np.asarray(ds.train_array)
        deg = np.bincount(ta[:, 1], minlength=ds.n_items)
        hot = np.argsort(-deg, kind='stable')
        user_rows = torch.zeros(n, dtype=torch.uint8, device=dev)
        user_rows[:ds.n_users] = 1
        out['graphs'][preset] = {'users': ds.n_users, 'items': ds.n_items, 'user_phase_gathers': int(len(ta)), 'nnz_A_hat': int(rowptr[-1])}
        variants.append((preset + '_user_phase_all', lambda csr=csr, x=x, y=y, m=user_rows: ops.spmm(csr, x, out=y, row_mask=m), int(len(ta)), 0))
        for H in HS:
            cm = torch.ones(n, dtype=torch.uint8, device=dev)
            cm[torch.from_numpy(ds.n_users + hot[:H]).to(dev)] = 0
            bits = ops.pack_mask_bits(cm)
            served = int(deg[hot[:H]].sum())
            variants.append(('%s_user_phase_without_top_%d' % (preset, H),
                             lambda csr=csr, x=x, y=y, m=user_rows, b=bits: ops.spmm(csr, x, out=y, row_mask=m, col_mask=b), int(len(ta)) - served, served))
            keep.append((cm, bits))
        # the whole launch, for scale
        variants.append((preset + '_whole_launch', lambda csr=csr, x=x, y=y: ops.spmm(csr, x, out=y), int(rowptr[-1]), 0))
        keep.append((csr, x, y, user_rows))
    out['order'] = [v[0] for v in variants]
    for name, fn, gathers, masked in variants:
        for _ in range(3):
            fn()                                             # warm (outside the counted dispatch block? no: counted, the summariser drops them)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(N_LAUNCH + 1)]
        e[0].record()
        for j in range(N_LAUNCH):
            fn()
            e[j + 1].record()
        torch.cuda.synchronize()
        out['launches'][name] = {'gathers': gathers, 'gathers_masked_out': masked, 'ms': [e[j].elapsed_time(e[j + 1]) for j in range(N_LAUNCH)]}
    print(json.dumps(out))


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True))
    return hits[0] if hits else None


def bench_line(path):
    for line in open(path):
        if line.startswith('{'):
            return json.loads(line)
    raise SystemExit('no JSON line in ' + path)


def summarize(tag):
    src = os.path.join(ROOT, 'gpurun_out', 'hot_' + tag)
    info = bench_line(os.path.join(src, 'pmc_l2.log'))
    timing = bench_line(os.path.join(src, 'plain.log'))
    per = N_LAUNCH + 3
    f = find(os.path.join(src, 'pmc_l2'), 'counter_collection.csv')
    vals = {}
    for counter in ('TCC_HIT_sum', 'TCC_MISS_sum'):
        rows = sorted((int(r['Dispatch_Id']), float(r['Counter_Value'])) for r in csv.DictReader(open(f))
                      if r['Counter_Name'] == counter and KERNEL in r['Kernel_Name'])
        if len(rows) != per * len(info['order']):
            raise SystemExit('%d dispatches of %s, expected %d' % (len(rows), KERNEL, per * len(info['order'])))
        vals[counter] = {name: [v for _, v in rows[j * per + 3:(j + 1) * per]] for j, name in enumerate(info['order'])}
    mean = lambda xs: sum(xs) / len(xs)
    med = lambda xs: sorted(xs)[len(xs) // 2]
    out = {'tag': tag, 'd': info['d'], 'kernel': 'spmm_csr_multirow_kernel<16,2,...> (row_mask = user rows; col_mask = all but the top-H items)',
           'how': 'TCC_HIT_sum / TCC_MISS_sum of one rocprofv3 --pmc pass, averages over %d launches per variant; times: HIP events of a run '
                  'without the profiler, median of %d; hot gathers = full launch - launch with the top-H item columns masked out' % (N_LAUNCH, N_LAUNCH),
           'graphs': {}}
    for preset, ginfo in info['graphs'].items():
        full = preset + '_user_phase_all'
        hit_f, miss_f = mean(vals['TCC_HIT_sum'][full]), mean(vals['TCC_MISS_sum'][full])
        ms_f = med(timing['launches'][full]['ms'])
        g = dict(ginfo)
        g['user_phase'] = {'ms': ms_f, 'l2_hits': hit_f, 'l2_misses': miss_f, 'l2_hit_rate': hit_f / (hit_f + miss_f)}
        g['whole_launch_ms'] = med(timing['launches'][preset + '_whole_launch']['ms'])
        g['top_H'] = {}
        for H in HS:
            name = '%s_user_phase_without_top_%d' % (preset, H)
            hit, miss = mean(vals['TCC_HIT_sum'][name]), mean(vals['TCC_MISS_sum'][name])
            ms = med(timing['launches'][name]['ms'])
            served = info['launches'][name]['gathers_masked_out']
            dh, dm = hit_f - hit, miss_f - miss
            g['top_H'][str(H)] = {
                'lds_KB': H * 256 / 1024, 'gathers': served, 'share_of_user_phase_gathers': served / ginfo['user_phase_gathers'],
                'l2_requests_of_the_hot_gathers': dh + dm, 'l2_requests_per_hot_gather': (dh + dm) / served,
                'share_never_reaching_l2 (2 lines per 256-B row expected)': 1.0 - (dh + dm) / (2.0 * served),
                'l2_hit_rate_of_the_hot_gathers': dh / (dh + dm) if dh + dm > 0 else None,
                'l2_misses_of_the_hot_gathers': dm, 'share_of_the_phase_l2_misses': dm / miss_f,
                'ms_without_them': ms, 'upper_bound_saving_ms': ms_f - ms, 'upper_bound_saving_of_user_phase': (ms_f - ms) / ms_f,
                'upper_bound_saving_of_whole_launch': (ms_f - ms) / g['whole_launch_ms']}
        out['graphs'][preset] = g
    path = os.path.join(ROOT, 'profiles', 'r06c_hot_item_rows_l2.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == 'summarize':
        summarize(sys.argv[2])
    else:
        run()

"""Condenses rocprofv3 CSV output (gpurun_out/prof_<tag>/) into profiles/<tag>_*.txt|json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True))
    return hits[0] if hits else None


def short(name):
    name = name.replace('igcn::', '').replace('(anonymous namespace)::', '')
    return name.split('(')[0][:90]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
    src = os.path.join('gpurun_out', 'prof_' + tag)
    os.makedirs('profiles', exist_ok=True)
    lines = []
    for sub, cmd, top in (('kt', 'bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-hbm-leg --no-config5  (the headline workload alone: '
                                 'the SpMM average below is that of the timed launches)', 8),
                          ('kt_full', 'bench.py --steps 100 --warmup 10 --no-cpu-baseline  (with eval, train step, HBM-bound leg, '
                                      'Gowalla-size step: the SpMM kernel name now also covers masked and small-graph launches)', 24)):
        ks = find(os.path.join(src, sub), 'kernel_stats.csv')
        if not ks:
            continue
        lines.append('# rocprofv3 --kernel-trace --stats -- python3 ' + cmd)
        lines.append('%-92s %8s %12s %10s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
        for j, r in enumerate(csv.DictReader(open(ks))):
            if j >= top:
                break
            lines.append('%-92s %8s %12.1f %10.2f %7s' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e3,
                                                          float(r['AverageNs']) / 1e3, r['Percentage']))
        lines.append('')
    pmc = {}
    for key, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write'), ('TCC_HIT_sum', 'pmc_l2'), ('TCC_MISS_sum', 'pmc_l2')):
        f = find(os.path.join(src, sub), 'counter_collection.csv')
        if not f:
            continue
        agg = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != key:
                continue
            a = agg[short(r['Kernel_Name'])]
            a[0] += float(r['Counter_Value']); a[1] += 1
        pmc[key] = {k: (v[0] / v[1], v[1]) for k, v in agg.items()}
    if pmc:
        lines.append('')
        lines.append('# rocprofv3 --pmc <counter> (separate passes), averages per dispatch')
        kernels = sorted({k for c in pmc.values() for k in c})
        for k in kernels:
            row = ['%-70s' % k]
            for c in ('FETCH_SIZE', 'WRITE_SIZE', 'TCC_HIT_sum', 'TCC_MISS_sum'):
                if c in pmc and k in pmc[c]:
                    row.append('%s=%.1f (n=%d)' % (c, pmc[c][k][0], pmc[c][k][1]))
            lines.append('  '.join(row))
    open(os.path.join('profiles', tag + '_rocprof_summary.txt'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    # HBM traffic per launch of the dominant kernel, corrected as MI355X_MICROARCH.md (HBM) prescribes:
    # FETCH_SIZE is in KiB and reads exactly half of a wide coalesced stream on gfx950 -> x2; WRITE_SIZE exact.
    spmm = [k for k in pmc.get('FETCH_SIZE', {}) if 'spmm_csr_multirow_kernel<16' in k or 'spmm_csr_rows_kernel<16' in k]
    if spmm:
        k = spmm[0]
        fetch_kib = pmc['FETCH_SIZE'][k][0]
        write_kib = pmc.get('WRITE_SIZE', {}).get(k, (0, 0))[0]
        cfg = {}
        try:                                   # the bench line of the FETCH_SIZE pass names the workload the counters belong to
            for line in open(os.path.join(src, 'pmc_fetch.log')):
                if line.startswith('{'):
                    cfg = json.loads(line).get('config', {})
        except Exception:
            pass
        hit = pmc.get('TCC_HIT_sum', {}).get(k, (0, 0))[0]
        miss = pmc.get('TCC_MISS_sum', {}).get(k, (0, 0))[0]
        out = {'tag': tag, 'kernel': k, 'preset': cfg.get('preset'), 'nnz': cfg.get('nnz'), 'd': cfg.get('d'),
               'l2_hit_rate': hit / (hit + miss) if hit + miss else None,
               'FETCH_SIZE_KiB_raw': fetch_kib, 'WRITE_SIZE_KiB_raw': write_kib,
               'spmm_hbm_bytes_per_launch': int(2 * fetch_kib * 1024 + write_kib * 1024),
               'correction': 'read bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-B requests at 64 B), '
                             'write bytes = WRITE_SIZE x 1024; Infinity-Cache hits are included in FETCH_SIZE'}
        json.dump(out, open(os.path.join('profiles', 'pmc_traffic.json'), 'w'), indent=1)
        print(out)


if __name__ == '__main__':
    main()

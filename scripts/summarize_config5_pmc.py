"""gpurun_out/prof5_<tag>/ (scripts/profile_config5.sh) -> profiles/<tag>_config5_rocprof_summary.txt and
profiles/pmc_traffic_config5.json (what bench.py's hbm_bound_leg reads).  Bytes per launch = 2 x FETCH_SIZE x 1024 +
WRITE_SIZE x 1024: FETCH_SIZE is in KiB and tallies the 128-byte requests of a wide read at 64 bytes on gfx950
(MI355X_MICROARCH.md, HBM); it counts what leaves L2 — Infinity-Cache hits included."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNEL = 'spmm_csr_rows_kernel<32'


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True))
    return hits[0] if hits else None


def bench_line(path):
    for line in open(path):
        if line.startswith('{'):
            return json.loads(line)
    raise SystemExit('no JSON line in ' + path)


def per_variant(csv_path, counter, info):
    """Counter values of the main SpMM kernel's dispatches in dispatch order, cut into the variants of the script's order."""
    rows = [(int(r['Dispatch_Id']), float(r['Counter_Value'])) for r in csv.DictReader(open(csv_path))
            if r['Counter_Name'] == counter and KERNEL in r['Kernel_Name']]
    rows.sort()
    n = info['n_launch']
    if len(rows) != n * len(info['order']):
        raise SystemExit('%s: %d dispatches of %s, expected %d' % (csv_path, len(rows), KERNEL, n * len(info['order'])))
    return {name: [v for _, v in rows[j * n + 1:(j + 1) * n]] for j, name in enumerate(info['order'])}      # first of each: cold


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
    src = os.path.join('gpurun_out', 'prof5_' + tag)
    info = bench_line(os.path.join(src, 'pmc_fetch.log'))
    timing = bench_line(os.path.join(src, 'kt.log'))
    vals = {}
    for counter, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write'), ('TCC_HIT_sum', 'pmc_l2'), ('TCC_MISS_sum', 'pmc_l2')):
        f = find(os.path.join(src, sub), 'counter_collection.csv')
        if not f:
            raise SystemExit('no counter csv under ' + sub)
        vals[counter] = per_variant(f, counter, info)
    mean = lambda xs: sum(xs) / len(xs)
    out = {'tag': tag, 'kernel': 'void spmm_csr_rows_kernel<32, false, false, false>', 'd': info['d'], 'world': info['world'], 'rank': info['rank'],
           'correction': 'bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE x 1024; what leaves L2, '
                         'Infinity-Cache hits included; averages over launches 2..%d of each variant' % info['n_launch'],
           'launches': {}}
    lines = ['# rocprofv3 of scripts/dev_config5_pmc.py (BASELINE config 5, rank %d of %d, d = %d): --kernel-trace --stats, then three --pmc passes'
             % (info['rank'], info['world'], info['d']), '']
    ks = find(os.path.join(src, 'kt'), 'kernel_stats.csv')
    if ks:
        lines.append('%-80s %8s %12s %10s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
        for j, r in enumerate(csv.DictReader(open(ks))):
            if j >= 6:
                break
            lines.append('%-80s %8s %12.1f %10.2f %7s' % (r['Name'].split('(')[0][:80], r['Calls'], float(r['TotalDurationNs']) / 1e3,
                                                          float(r['AverageNs']) / 1e3, r['Percentage']))
        lines.append('')
    lines.append('%-12s %10s %12s %9s %14s %14s %12s %9s %10s %10s' % ('launch', 'rows', 'nnz', 'ms', 'FETCH_KiB_raw', 'WRITE_KiB_raw',
                                                                      'bytes(2F+W)', 'GB/s', 'of 8 TB/s', 'L2 hit'))
    for name in info['order']:
        fk, wk = mean(vals['FETCH_SIZE'][name]), mean(vals['WRITE_SIZE'][name])
        hit, miss = mean(vals['TCC_HIT_sum'][name]), mean(vals['TCC_MISS_sum'][name])
        nbytes = int(2 * fk * 1024 + wk * 1024)
        ms = mean(timing['launches'][name]['ms'][1:])                  # HIP events of the kernel-trace run (no counters)
        L = info['launches'][name]
        out['launches'][name] = {'rows': L['rows'], 'nnz': L['nnz'], 'FETCH_SIZE_KiB_raw': fk, 'WRITE_SIZE_KiB_raw': wk, 'bytes': nbytes,
                                 'l2_hit_rate': hit / (hit + miss) if hit + miss else None, 'ms_in_profile_run': ms,
                                 'GBps_in_profile_run': nbytes / ms / 1e6, 'algorithmic_bytes': L['algorithmic_bytes'],
                                 'index_and_output_bytes': L['index_and_output_bytes']}
        lines.append('%-12s %10d %12d %9.3f %14.0f %14.0f %12d %9.0f %10.3f %10.3f' % (name, L['rows'], L['nnz'], ms, fk, wk, nbytes, nbytes / ms / 1e6,
                                                                                     nbytes / ms / 1e6 / 8000., hit / (hit + miss)))
    lines.append('')
    lines.append('algorithmic bytes (SURVEY 8(d): nnz*(8+4d) + rows*(4d+4)): ' +
                 ', '.join('%s %d' % (n, info['launches'][n]['algorithmic_bytes']) for n in info['order']))
    os.makedirs('profiles', exist_ok=True)
    open(os.path.join('profiles', tag + '_config5_rocprof_summary.txt'), 'w').write('\n'.join(lines) + '\n')
    json.dump(out, open(os.path.join('profiles', 'pmc_traffic_config5.json'), 'w'), indent=1)
    print('\n'.join(lines))


if __name__ == '__main__':
    main()

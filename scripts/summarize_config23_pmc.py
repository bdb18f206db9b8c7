"""gpurun_out/prof23_<tag>/ (scripts/profile_config23.sh) -> profiles/<tag>_config23_rocprof_summary.txt and
profiles/pmc_traffic_config23.json (what bench.py's propagation lines of configs 2 and 3 read).  Bytes per launch =
2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024: FETCH_SIZE is in KiB and tallies the 128-byte requests of a wide read at 64 bytes on
gfx950 (MI355X_MICROARCH.md, HBM); it counts what leaves L2 — Infinity-Cache hits included."""
import csv
import glob
import json
import os
import sys

KERNEL = 'spmm_csr_multirow_kernel<16'


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True))
    return hits[0] if hits else None


def bench_line(path):
    for line in open(path):
        if line.startswith('{'):
            return json.loads(line)
    raise SystemExit('no JSON line in ' + path)


def per_variant(csv_path, counter, info):
    rows = [(int(r['Dispatch_Id']), float(r['Counter_Value'])) for r in csv.DictReader(open(csv_path))
            if r['Counter_Name'] == counter and KERNEL in r['Kernel_Name']]
    rows.sort()
    n = info['n_launch']
    setup = len(rows) - n * len(info['order'])                  # (the script's set-up launches come first)
    if setup < 0:
        raise SystemExit('%s: %d dispatches of %s, expected at least %d' % (csv_path, len(rows), KERNEL, n * len(info['order'])))
    rows = rows[setup:]
    return {name: [v for _, v in rows[j * n + 1:(j + 1) * n]] for j, name in enumerate(info['order'])}      # first of each: cold


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
    src = os.path.join('gpurun_out', 'prof23_' + tag)
    info = bench_line(os.path.join(src, 'pmc_fetch.log'))
    timing = bench_line(os.path.join(src, 'kt.log'))
    vals = {}
    for counter, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write'), ('TCC_HIT_sum', 'pmc_l2'), ('TCC_MISS_sum', 'pmc_l2')):
        f = find(os.path.join(src, sub), 'counter_collection.csv')
        if not f:
            raise SystemExit('no counter csv under ' + sub)
        vals[counter] = per_variant(f, counter, info)
    mean = lambda xs: sum(xs) / len(xs)
    out = {'tag': tag, 'kernel': 'spmm_csr_multirow_kernel<16,2,*>', 'd': info['d'], 'gowalla': info['gowalla'], 'yelp': info['yelp'],
           'correction': 'bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE x 1024; what leaves L2, '
                         'Infinity-Cache hits included; main kernel only (the long-row reduce kernel is not in these counters); averages '
                         'over launches 2..%d of each variant' % info['n_launch'],
           'launches': {}}
    lines = ['# rocprofv3 of scripts/dev_r05_config23_pmc.py (BASELINE configs 2 and 3, d = %d): --kernel-trace --stats, then three --pmc passes' % info['d'], '']
    ks = find(os.path.join(src, 'kt'), 'kernel_stats.csv')
    if ks:
        lines.append('%-80s %8s %12s %10s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
        for j, r in enumerate(csv.DictReader(open(ks))):
            if j >= 8:
                break
            lines.append('%-80s %8s %12.1f %10.2f %7s' % (r['Name'].split('(')[0][:80], r['Calls'], float(r['TotalDurationNs']) / 1e3,
                                                          float(r['AverageNs']) / 1e3, r['Percentage']))
        lines.append('')
    lines.append('%-36s %9s %10s %8s %13s %13s %12s %8s %9s %7s' % ('launch', 'rows', 'nnz', 'ms', 'algorithmic_B', 'bytes(2F+W)', 'GB/s beyond L2',
                                                                   'alg GB/s', 'traffic/alg', 'L2 hit'))
    for name in info['order']:
        fk, wk = mean(vals['FETCH_SIZE'][name]), mean(vals['WRITE_SIZE'][name])
        hit, miss = mean(vals['TCC_HIT_sum'][name]), mean(vals['TCC_MISS_sum'][name])
        nbytes = int(2 * fk * 1024 + wk * 1024)
        ms = mean(timing['launches'][name]['ms'][1:])                  # HIP events of the kernel-trace run (no counters); main + reduce kernel
        L = info['launches'][name]
        out['launches'][name] = {'rows': L['rows'], 'nnz': L['nnz'], 'FETCH_SIZE_KiB_raw': fk, 'WRITE_SIZE_KiB_raw': wk, 'bytes': nbytes,
                                 'l2_hit_rate': hit / (hit + miss) if hit + miss else None, 'ms_in_profile_run': ms,
                                 'GBps_in_profile_run': nbytes / ms / 1e6, 'algorithmic_bytes': L['algorithmic_bytes']}
        lines.append('%-36s %9d %10d %8.4f %13d %13d %12.0f %8.0f %9.2f %7.3f' % (name, L['rows'], L['nnz'], ms, L['algorithmic_bytes'], nbytes,
                                                                                  nbytes / ms / 1e6, L['algorithmic_bytes'] / ms / 1e6,
                                                                                  nbytes / L['algorithmic_bytes'], hit / (hit + miss)))
    os.makedirs('profiles', exist_ok=True)
    open(os.path.join('profiles', tag + '_config23_rocprof_summary.txt'), 'w').write('\n'.join(lines) + '\n')
    json.dump(out, open(os.path.join('profiles', 'pmc_traffic_config23.json'), 'w'), indent=1)
    print('\n'.join(lines))


if __name__ == '__main__':
    main()

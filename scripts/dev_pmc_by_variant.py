"""Per-variant averages of rocprofv3 --pmc passes of a script that launches N variants K times each in a fixed order
(scripts/dev_xcd_plan_ab.py with PMC=1).  usage: dev_pmc_by_variant.py <dir-with-pass-subdirs> <log-with-order-json> <kernel-substr>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, log, kern = sys.argv[1], sys.argv[2], sys.argv[3]
order = None
for line in open(log):
    if line.startswith('{') and 'pmc_order' in line:
        order = json.loads(line)
names, per = order['pmc_order'], order['launches_per_variant']
res = defaultdict(dict)
for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
    rows = defaultdict(dict)                    # dispatch id -> counter -> value
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            rows[int(r['Dispatch_Id'])][r['Counter_Name']] = rows[int(r['Dispatch_Id'])].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    ids = sorted(rows)
    if len(ids) > len(names) * per:                         # launches before the measured sequence (a script's own checks): the LAST ones count
        ids = ids[-len(names) * per:]
    if len(ids) != len(names) * per:
        print('skip', f, len(ids), 'dispatches, expected', len(names) * per, file=sys.stderr)
        continue
    for i, name in enumerate(names):
        chunk = ids[i * per + 1:(i + 1) * per]              # drop the first launch of a variant (cold)
        for c in rows[chunk[0]]:
            res[name][c] = sum(rows[j][c] for j in chunk) / len(chunk)
for name in names:
    r = res[name]
    line = {'variant': name}
    line.update({k: round(v, 1) for k, v in r.items()})
    if 'TCC_HIT_sum' in r and 'TCC_MISS_sum' in r:
        line['l2_hit_rate'] = round(r['TCC_HIT_sum'] / (r['TCC_HIT_sum'] + r['TCC_MISS_sum']), 4)
    if 'FETCH_SIZE' in r:
        line['fabric_MB'] = round((2 * r['FETCH_SIZE'] + r.get('WRITE_SIZE', 0.0)) * 1024 / 1e6, 1)
    print(json.dumps(line))

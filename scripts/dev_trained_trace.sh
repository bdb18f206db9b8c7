#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/trained_trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/scripts/dev_trained_trace.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('$O/t/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'topk_row_stats' in r['Kernel_Name']]
i0 = idx[-1]                                   # the last call: from the two tables' statistics on
t0 = int(rows[i0]['Start_Timestamp']); prev = t0
for r in rows[i0:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f  gap %7.1f  dur %8.1f  %s' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:90]))
    prev = e
PY

#!/bin/bash
# Developer: kernel-by-kernel timeline of ONE full evaluation (recommend_all, Amazon-like): start offsets, durations, gaps
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/eval_timeline; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/scripts/dev_eval_times.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('$O/t/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the 6th evaluation of the first 'auto' block: find starts of propagation (first spmm kernel after a gap > 200 us)
idx = [i for i, r in enumerate(rows) if 'spmm_csr_multirow' in r['Kernel_Name']]      # three per evaluation
i0, i1 = idx[15], idx[18]                                                               # the sixth evaluation
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f  gap %7.1f  dur %8.1f  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:90]))
    prev_end = e
print('total us', (prev_end - t0) / 1e3)
PY

#!/bin/bash
# Developer: the candidate-sweep kernel's own time in every ablation build (rocprofv3 kernel trace; the ablated builds return
# garbage lists, so whole-call times would include a fall-back for every user).  D=64|128 WIDE=0|1 bash scripts/dev_topk_variants_trace.sh [variants]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/topk_variants_d${D:-64}_w${WIDE:-x}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TOPK_MODE=fast ONLY_K20=1 PYTHONPATH=$R
for v in ${@:-base nohits noselect noloada bare sametile noflush}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 $R/scripts/dev_topk_variants.py run $v > $O/$v.log 2>&1 || { tail -5 $O/$v.log; echo "$v failed"; continue; }
  python3 - <<PY
import csv, glob
f = glob.glob('$O/$v/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'score_topk_kernel' in r['Name'] and ', 2, false' in r['Name']:
        print('%-14s %-60s calls %3s avg %9.1f us  min %9.1f us' % ('$v', r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done

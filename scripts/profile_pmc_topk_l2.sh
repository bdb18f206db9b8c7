#!/bin/bash
# L2 hits / misses / bytes beyond L2 of the top-k kernels (separate rocprofv3 --pmc passes).  D=64|128 TOPK_MODE=fast|exact bash scripts/profile_pmc_topk_l2.sh <tag>
set -o pipefail
TAG=${1:-l2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_topk_l2_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
i=0
for P in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $REPO/scripts/dev_topk_once.py > $OUT/p$i.log 2>&1 || { tail -3 $OUT/p$i.log; echo "pass $i failed"; }
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'score_topk_kernel' in r['Kernel_Name'] or 'rescore' in r['Kernel_Name']:
            a = agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for kern in sorted(agg):
    print(kern)
    for k in sorted(agg[kern]):
        print('   %-32s %16.0f   (avg per dispatch, n=%d)' % (k, agg[kern][k][0] / agg[kern][k][1], agg[kern][k][1]))
PY

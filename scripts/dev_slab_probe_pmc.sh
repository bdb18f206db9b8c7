#!/bin/bash
# L2 hits / misses / bytes beyond L2 of the slab probe, one piece width per run (separate --pmc passes)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/slab_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
for piece in 64 32 16 8; do
  i=0
  for P in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $P --output-format csv -d $OUT/w${piece}_p$i -- python3 $REPO/scripts/dev_slab_probe.py amazon $piece > $OUT/w${piece}_p$i.log 2>&1 || { tail -5 $OUT/w${piece}_p$i.log; echo "pass $piece/$i failed"; }
  done
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('$OUT/w*/**/*counter_collection.csv', recursive=True):
    w = re.search(r'/w(\d+)_p', f).group(1)
    for r in csv.DictReader(open(f)):
        if 'slab_gather_kernel' in r['Kernel_Name']:
            a = agg[w][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for w in sorted(agg, key=int):
    print('piece floats', w)
    for k in sorted(agg[w]):
        print('   %-28s %16.0f   (avg per dispatch, n=%d)' % (k, agg[w][k][0] / agg[w][k][1], agg[w][k][1]))
PY

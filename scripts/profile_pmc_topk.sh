#!/bin/bash
# SQ counters of score_topk_kernel only (separate rocprofv3 --pmc passes). Usage: bash scripts/profile_pmc_topk.sh <tag>
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_topk_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM"
P3="GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $REPO/scripts/dev_topk_once.py > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; echo "pass $i failed"; }
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'score_topk_kernel' in r['Kernel_Name']:
            a = agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for kern in sorted(agg):
    print(kern)
    for k in sorted(agg[kern]):
        print('   %-32s %16.0f   (avg per dispatch, n=%d)' % (k, agg[kern][k][0] / agg[kern][k][1], agg[kern][k][1]))
PY

#!/bin/bash
# SQ-level counters for the two hot kernels (separate rocprofv3 --pmc passes). Usage: bash scripts/profile_pmc.sh <tag>
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE"
P3="GRBM_GUI_ACTIVE GRBM_COUNT"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/topk_p$i -- python3 $REPO/scripts/dev_topk_once.py > $OUT/topk_p$i.log 2>&1 || { tail -5 $OUT/topk_p$i.log; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/spmm_p$i -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-hbm-leg --no-config5 > $OUT/spmm_p$i.log 2>&1 || { tail -5 $OUT/spmm_p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for sub, kern in (('topk', 'score_topk_kernel'), ('spmm', 'spmm_csr_')):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob('$OUT/%s_p*/**/*counter_collection.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    print(sub, kern)
    for k in sorted(agg):
        print('   %-28s %16.0f   (avg per dispatch, n=%d)' % (k, agg[k][0] / agg[k][1], agg[k][1]))
PY

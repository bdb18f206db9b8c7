#!/bin/bash
# scripts/dev_r06_hot_rows_l2.py once plainly (HIP-event times) and once under rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum (the
# program directly after `--`).  Usage (on the GPU box): bash scripts/dev_r06_hot_rows_l2.sh <tag>
set -o pipefail
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/hot_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROG="$REPO/scripts/dev_r06_hot_rows_l2.py"
timeout -k 10 300 python3 $PROG > $OUT/plain.log 2>$OUT/plain.err || { tail -5 $OUT/plain.err; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex 'spmm_' --output-format csv -d $OUT/pmc_l2 -- python3 $PROG > $OUT/pmc_l2.log 2>$OUT/pmc_l2.err || { tail -5 $OUT/pmc_l2.err; exit 1; }
find $OUT -name '*counter_collection.csv' | head

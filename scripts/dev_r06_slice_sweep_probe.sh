#!/bin/bash
# scripts/dev_r06_slice_sweep_probe.py plainly (HIP-event times) and under two rocprofv3 --pmc passes (FETCH_SIZE; TCC_HIT/MISS), the
# program directly after `--`.  Usage (on the GPU box): bash scripts/dev_r06_slice_sweep_probe.sh <tag>
set -o pipefail
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/slice_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROG="$REPO/scripts/dev_r06_slice_sweep_probe.py"
timeout -k 10 300 python3 $PROG > $OUT/plain.log 2>$OUT/plain.err || { tail -5 $OUT/plain.err; exit 1; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'spmm_csr' --output-format csv -d $OUT/pmc_fetch -- python3 $PROG > $OUT/pmc_fetch.log 2>$OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex 'spmm_csr' --output-format csv -d $OUT/pmc_l2 -- python3 $PROG > $OUT/pmc_l2.log 2>$OUT/pmc_l2.err || { tail -5 $OUT/pmc_l2.err; exit 1; }
cat $OUT/plain.log

#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# Usage: bash scripts/profile.sh <tag>
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# pass 1: the headline workload alone (the timed region + the in-run gather roof), so that the SpMM kernel's average is
# that of the launches bench.py times; pass 2: everything (eval, train step, HBM-bound leg, small-graph step)
ARGS="$REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-hbm-leg --no-config5"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ARGS > $OUT/kt.log 2>&1 || exit 1
ARGSF="$REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-config5"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_full -- python3 $ARGSF > $OUT/kt_full.log 2>&1 || exit 1
ARGS2="$REPO/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-hbm-leg --no-config5"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS2 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS2 > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ARGS2 > $OUT/pmc_l2.log 2>&1 || exit 1
find $OUT -name '*.csv' | head -30

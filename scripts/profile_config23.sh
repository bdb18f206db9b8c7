#!/bin/bash
# Counters under the propagation launches of BASELINE configs 2 and 3 (scripts/dev_r05_config23_pmc.py): rocprofv3 kernel trace + three
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC_HIT/MISS), restricted to the SpMM kernels; the program directly after `--`.
# Usage (on the GPU box): bash scripts/profile_config23.sh <tag>   then   python scripts/summarize_config23_pmc.py <tag>
set -o pipefail
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof23_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROG="$REPO/scripts/dev_r05_config23_pmc.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $PROG > $OUT/kt.log 2>$OUT/kt.err || { tail -5 $OUT/kt.err; exit 1; }
for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  NAME=${P%%:*}; CTRS=${P#*:}
  timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-include-regex 'spmm_' --output-format csv -d $OUT/pmc_$NAME -- python3 $PROG > $OUT/pmc_$NAME.log 2>$OUT/pmc_$NAME.err || { tail -5 $OUT/pmc_$NAME.err; exit 1; }
done
find $OUT -name '*.csv' | head -20

#!/bin/bash
# Same-process timings of scripts/dev_community_ab.py, then L2 / fabric counters per variant (separate --pmc passes).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/community_ab; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/dev_community_ab.py > $O/timing.jsonl 2>$O/timing.err || { tail -5 $O/timing.err; exit 1; }
cat $O/timing.jsonl
for P in "l2:TCC_HIT_sum TCC_MISS_sum" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  NAME=${P%%:*}; CTRS=${P#*:}
  PMC=1 timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-include-regex 'spmm_csr_multirow' --output-format csv -d $O/pmc/$NAME -- python3 $R/scripts/dev_community_ab.py > $O/pmc_$NAME.log 2>$O/pmc_$NAME.err || { tail -5 $O/pmc_$NAME.err; exit 1; }
done
python3 $R/scripts/dev_pmc_by_variant.py $O/pmc $O/pmc_l2.log spmm_csr_multirow | tee $O/pmc_by_variant.jsonl

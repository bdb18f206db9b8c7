set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R && timeout -k 10 500 python scripts/dev_xcd_plan_ab.py > $O/ab.json 2> $O/ab.err || { tail -20 $O/ab.err; exit 1; }
cat $O/ab.json
cd /tmp && export TMPDIR=/tmp
PMC=1 timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 $R/scripts/dev_xcd_plan_ab.py > $O/pmc_l2.log 2>&1 || { tail -5 $O/pmc_l2.log; exit 1; }
PMC=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/pmc_fw -- python3 $R/scripts/dev_xcd_plan_ab.py > $O/pmc_fw.log 2>&1 || { tail -5 $O/pmc_fw.log; exit 1; }
cd $R && python scripts/dev_pmc_by_variant.py $O $O/pmc_l2.log spmm_csr_multirow > $O/pmc_by_variant.jsonl; cat $O/pmc_by_variant.jsonl
find $O -name '*.csv' -size +20M -delete

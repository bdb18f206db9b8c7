set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 500 python scripts/dev_config5_shares.py > $O/r03f_config5_shares.json 2> $O/r03f_config5.err || { tail -20 $O/r03f_config5.err; exit 1; }
python -c "
import json; d=json.load(open('$O/r03f_config5_shares.json')); print(json.dumps({k:v for k,v in d.items() if k!='workload'})[:3000])"
timeout -k 10 500 python -m pytest tests/test_spmm_gpu.py -q -x -k "config5 or xcd" > $O/r03f_tests.log 2>&1 || { tail -30 $O/r03f_tests.log; exit 1; }
tail -3 $O/r03f_tests.log
timeout -k 10 600 python bench.py > $O/r03f_bench.json 2> $O/r03f_bench.err || { tail -30 $O/r03f_bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/r03f_bench.json')); e=d.pop('extras'); print(json.dumps(d)[:6000])"

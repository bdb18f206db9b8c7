set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_score_bpr_gpu.py tests/test_fuzz_gpu.py -x -q > $O/r03j_tests.log 2>&1 || { tail -40 $O/r03j_tests.log; exit 1; }
tail -3 $O/r03j_tests.log
timeout -k 10 300 python scripts/dev_topk_item_order.py 2>&1 | grep -v amdgpu.ids
bash scripts/dev_topk_trace.sh

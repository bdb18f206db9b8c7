# Developer: kernel-by-kernel times of the two-stage evaluation (rocprofv3 --kernel-trace --stats), Amazon-like, k = 20
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/topk_trace_d${D:-64}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TOPK_MODE=fast timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast -- python3 $R/scripts/dev_topk_once.py 5 > $O/fast.log 2>&1 || { tail -5 $O/fast.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('$O/fast/**/*kernel_stats.csv', recursive=True)[0]
for j, r in enumerate(csv.DictReader(open(f))):
    if j < 14: print('%-100s %6s %10.1f us avg' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3))
PY

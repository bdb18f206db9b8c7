#!/bin/bash
# Developer: GPU-busy time vs wall time of a LightGCN training step (how much is launch gaps?)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/step_gaps_${PRESET:-amazon}_${INDEX:-1}; rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $REPO/scripts/dev_step_profile.py ${PRESET:-amazon} ${INDEX:-1} > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
# last 40 steps: find bpr_sample_kernel launches as step markers
marks = [i for i, r in enumerate(rows) if 'bpr_sample_kernel' in r[2]]
a, b = marks[-41], marks[-1]
seg = rows[a:b]
wall = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
print('steps 40  wall_us/step %.1f  gpu_busy_us/step %.1f  kernels/step %.1f  idle %.1f%%' % (wall / 40e3, busy / 40e3, len(seg) / 40, 100 * (1 - busy / wall)))
import collections
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    k = n.split('(')[0].replace('void ', '').replace('igcn::', '')[:90]
    agg[k][0] += e - s; agg[k][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print('  %-62s %7.1f us/step  %5.1f calls/step' % (k, v[0] / 40e3, v[1] / 40))
print('--- one step: start_us  duration_us  kernel')
t0 = rows[marks[-2]][0]
for s_, e_, n_ in rows[marks[-2]:marks[-1]]:
    print('%8.1f %7.1f  %s' % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, n_.replace('void ', '').replace('igcn::', '').replace('at::native::', '')[:80]))
PY

#!/bin/bash
# Runs the given steps ("name|seconds|command" each) one after the other on the GPU box; a step that fails goes on to the
# next one, a step that is KILLED AT ITS LIMIT (or by a signal) stops the sequence — no GPU step is started after a hang.
# Output of every step: gpurun_out/<name>.log.  Usage: bash scripts/gpu_steps.sh "tests|600|python -m pytest ..." ...
mkdir -p gpurun_out
overall=0
for step in "$@"; do
  name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 $secs bash -c "$cmd" > gpurun_out/$name.log 2>&1
  rc=$?
  echo "== $name rc=$rc"; tail -n 6 gpurun_out/$name.log
  if [ $rc -eq 124 ] || [ $rc -ge 129 ]; then echo "== $name was killed: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && overall=$rc
done
exit $overall

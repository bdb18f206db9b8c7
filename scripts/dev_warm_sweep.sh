#!/bin/bash
# Developer: scripts/dev_warm_sweep.py with the warm-up lengths / epochs given as arguments:  bash scripts/dev_warm_sweep.sh 0,6,12 3
cd "$(dirname "$0")/.."
export WARMS="${1:-0,32,64,128,256,512}" EPOCHS="${2:-2}"
exec python scripts/dev_warm_sweep.py

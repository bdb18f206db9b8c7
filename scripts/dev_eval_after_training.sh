#!/bin/bash
# Developer: the full evaluation timed call by call, as bench.py does it, after 35 training steps (bench.py's order of legs) and without
cd "$(dirname "$0")/.."
python scripts/dev_eval_knob_ab.py topk_fast_warm -1 0 || exit 1
TRAIN_STEPS=35 python scripts/dev_eval_knob_ab.py topk_fast_warm -1 0

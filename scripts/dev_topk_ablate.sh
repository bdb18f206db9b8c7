#!/bin/bash
# Developer: the two-stage evaluation under the ablation builds of scripts/dev_topk_variants.py (built here beforehand), each with the
# default plan and with the staging depth held at its 8-waves-per-CU value ("topk_cap" 8)
cd "$(dirname "$0")/.."
export TOPK_MODE=fast ONLY_K20=1 TUNES=';topk_cap=8'
exec python scripts/dev_topk_variants.py run "$@"
